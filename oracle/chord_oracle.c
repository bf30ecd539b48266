/*
 * chord_oracle.c — CPU oracle for PSF-Attn's chord-sparse matmul path.
 *
 * TEST INFRASTRUCTURE ONLY. This is a plain-C restatement of the reference algorithm used as the CHECKER:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. The product path
 * (sparsefactorization_amd/) never imports, links or calls anything in oracle/ and fails loudly without
 * its HIP library.
 *
 * What is pinned and by what (see DESIGN.md "Oracle"):
 *   - index pattern: oracle_chord_indices is checked element for element against the output of the
 *     reference's own get_chord_indices_assym, run in the build container from /root/reference
 *     (fixtures tests/golden/chord_indices_*.npz, generator oracle/gen_golden.py);
 *   - spmm arithmetic: torch_sparse.spmm lives in the un-vendored dependency torch-sparse==0.6.11
 *     (requirements.txt:146), which is not in /root/reference and not installed. It is restated from its
 *     published algorithm (index_select -> mul -> scatter_add) and cross-checked against the reference's
 *     own in-repo statement of the same operator, spmul/spmul_cuda.cu:24,79-80,105-108 (oracle_spmul_*);
 *     the reference holds no test, golden vector or known-answer value for this path, so
 *     the arithmetic itself is pinned only by outputs of the reference's PSFNet.forward run here with that
 *     restated spmm injected (tests/golden/psfnet_*.npz) — stated as such, not as reference test vectors;
 *   - spmul/spmul_cuda.cu cannot be built here (CUDA + torch extension, no nvcc): there is no oracle/_ref.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no -ffast-math, no -march=native)
 */
#include <stdint.h>
#include <stddef.h>

/*
 * get_chord_indices_assym(n_vec, n_link) — SyntheticExperiments/psf.py:7-32 (identical copies:
 * LRA/psf.py:7-32, Genome_Clf/psf.py:7-32, attention_block.py:14-39):
 *     rows = [i for each of n_link] for i in range(n_vec)
 *     cols = [i] + [(i + 2**k) % n_vec for k in range(n_link - 1)] for i in range(n_vec)
 * Python integers do not overflow, so 2**k is reduced mod n_vec by repeated doubling here.
 * Duplicated links are kept, as in the reference. rows/cols hold n_vec*n_link entries each.
 */
int oracle_chord_indices(int64_t n_vec, int64_t n_link, int64_t* rows, int64_t* cols) {
  if (n_vec < 1 || n_link < 1) return -1;
  for (int64_t i = 0; i < n_vec; ++i) {
    int64_t* r = rows + i * n_link;
    int64_t* c = cols + i * n_link;
    for (int64_t j = 0; j < n_link; ++j) r[j] = i;
    c[0] = i;
    int64_t pw = 1 % n_vec; /* 2**0 mod n_vec */
    for (int64_t k = 0; k < n_link - 1; ++k) {
      c[k + 1] = (i + pw) % n_vec;
      pw = (pw * 2) % n_vec;
    }
  }
  return 0;
}

/* get_offsets(n_link_all) — spmul/spmul.py:8-9: [0] + [2**k for k in range(n_link_all - 1)], unreduced. */
int oracle_spmul_offsets(int64_t n_link_all, int64_t* offsets) {
  if (n_link_all < 1 || n_link_all > 63) return -1;
  offsets[0] = 0;
  for (int64_t k = 0; k < n_link_all - 1; ++k) offsets[k + 1] = (int64_t)1 << k;
  return 0;
}

#define T float
#define SUF _f32
#include "chord_oracle_impl.h"
#undef T
#undef SUF

#define T double
#define SUF _f64
#include "chord_oracle_impl.h"
#undef T
#undef SUF
