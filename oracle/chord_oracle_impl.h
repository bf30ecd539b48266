/*
 * chord_oracle_impl.h — type-generic bodies of the CPU oracle. Included twice by chord_oracle.c with
 * T / SUF defined (float/_f32, double/_f64). TEST INFRASTRUCTURE ONLY — see chord_oracle.c.
 *
 * Every sum is "rounded product, then rounded add" in the order the cited reference code visits the
 * terms; the file is compiled with -ffp-contract=off so no FMA is formed.
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/*
 * torch_sparse.spmm(index, value, m, n, matrix) — third-party torch-sparse==0.6.11 (requirements.txt:146),
 * absent from /root/reference; published algorithm (torch_sparse/spmm.py, 0.6.x):
 *     row, col = index
 *     out = matrix.index_select(-2, col)          # [.., nnz, C]
 *     out = out * value.unsqueeze(-1)             # value [B, nnz]
 *     out = scatter_add(out, row, dim=-2, dim_size=m)
 * Called at SyntheticExperiments/psf.py:178-184, LRA/psf.py:230-236, Genome_Clf/psf.py:220-226,
 * attention_block.py:164-170, LRA/attention_maps/pathfinder_inference.py:66-81, imdb_inference.py:45-59.
 * CPU scatter_add walks nnz in index order, so a row's terms are added in list order starting from 0.
 * Generic over the index list: nothing here knows about the chord pattern.
 *   value  [B, nnz];  matrix [B, n, C] (mat_bstride = n*C) or [n, C] (mat_bstride = 0);  out [B, m, C]
 */
int FN(oracle_spmm)(const int64_t* rows, const int64_t* cols, int64_t nnz, const T* value, int64_t B, int64_t m,
                    int64_t n, const T* matrix, int64_t mat_bstride, int64_t C, T* out) {
  for (int64_t j = 0; j < nnz; ++j)
    if (rows[j] < 0 || rows[j] >= m || cols[j] < 0 || cols[j] >= n) return -1;
  for (int64_t b = 0; b < B; ++b) {
    T* ob = out + b * m * C;
    const T* mb = matrix + b * mat_bstride;
    const T* vb = value + b * nnz;
    for (int64_t i = 0; i < m * C; ++i) ob[i] = (T)0;
    for (int64_t j = 0; j < nnz; ++j) {
      const T* src = mb + cols[j] * C;
      T* dst = ob + rows[j] * C;
      const T w = vb[j];
      for (int64_t c = 0; c < C; ++c) {
        const T prod = src[c] * w; /* out = out * value.unsqueeze(-1) */
        dst[c] = dst[c] + prod;    /* scatter_add */
      }
    }
  }
  return 0;
}

/*
 * spmul forward — spmul/spmul_cuda.cu:20-27:
 *     Z[i][p][d] += F[i][p][k] * V[i][(p + offsets[k]) % n_vec][d]      k = 0 .. n_link_all-1, Z starts at 0
 * offsets are arbitrary int64 (spmul/spmul.py:8-9 builds [0, 1, 2, 4, ...]); Python-style modulo.
 */
int FN(oracle_spmul_fwd)(const T* F, const T* V, int64_t v_bstride, const int64_t* offsets, int64_t B, int64_t N,
                         int64_t L, int64_t C, T* Z) {
  for (int64_t b = 0; b < B; ++b)
    for (int64_t p = 0; p < N; ++p)
      for (int64_t d = 0; d < C; ++d) {
        T z = (T)0;
        for (int64_t k = 0; k < L; ++k) {
          int64_t q = (p + offsets[k]) % N;
          if (q < 0) q += N;
          const T prod = F[(b * N + p) * L + k] * V[b * v_bstride + q * C + d];
          z = z + prod;
        }
        Z[(b * N + p) * C + d] = z;
      }
  return 0;
}

/*
 * spmul backward w.r.t. V — spmul/spmul_cuda.cu:75-84:
 *     j = (p - offsets[k] + n_vec) % n_vec;   dJdV[i][p][d] += F[i][j][k] * dJdZ[i][j][d]
 */
int FN(oracle_spmul_bwd_dv)(const T* dZ, const T* F, const int64_t* offsets, int64_t B, int64_t N, int64_t L,
                            int64_t C, T* dV) {
  for (int64_t b = 0; b < B; ++b)
    for (int64_t p = 0; p < N; ++p)
      for (int64_t d = 0; d < C; ++d) {
        T a = (T)0;
        for (int64_t k = 0; k < L; ++k) {
          int64_t j = (p - offsets[k]) % N;
          if (j < 0) j += N;
          const T prod = F[(b * N + j) * L + k] * dZ[(b * N + j) * C + d];
          a = a + prod;
        }
        dV[(b * N + p) * C + d] = a;
      }
  return 0;
}

/*
 * spmul backward w.r.t. F — spmul/spmul_cuda.cu:102-111:
 *     j = (p + offsets[k]) % n_vec;   dJdF[i][p][k] += dJdZ[i][p][d] * V[i][j][d]     d = 0 .. n_dim-1
 */
int FN(oracle_spmul_bwd_df)(const T* dZ, const T* V, int64_t v_bstride, const int64_t* offsets, int64_t B,
                            int64_t N, int64_t L, int64_t C, T* dF) {
  for (int64_t b = 0; b < B; ++b)
    for (int64_t p = 0; p < N; ++p)
      for (int64_t k = 0; k < L; ++k) {
        int64_t j = (p + offsets[k]) % N;
        if (j < 0) j += N;
        T a = (T)0;
        for (int64_t d = 0; d < C; ++d) {
          const T prod = dZ[(b * N + p) * C + d] * V[b * v_bstride + j * C + d];
          a = a + prod;
        }
        dF[(b * N + p) * L + k] = a;
      }
  return 0;
}

/*
 * The hot loop of PSFNet.forward — SyntheticExperiments/psf.py:167-188 (LRA/psf.py:220-240):
 *     res_conn = V;  for m < n_W:  V = spmm(chord_indicies, W_m.reshape(B, N*L), N, N, V);  V = V + res_conn
 * W_all [M, B, N*L]; V0 [B, N, C]; steps_out [M, B, N, C] receives V after every step (last = result).
 * Uses the generic COO spmm above with the caller's index list.
 */
int FN(oracle_chain)(const int64_t* rows, const int64_t* cols, int64_t nnz, const T* W_all, int64_t M,
                     const T* V0, int64_t B, int64_t N, int64_t C, int use_residual, T* steps_out) {
  const int64_t sz = B * N * C;
  for (int64_t m = 0; m < M; ++m) {
    const T* in = m == 0 ? V0 : steps_out + (m - 1) * sz;
    T* o = steps_out + m * sz;
    int rc = FN(oracle_spmm)(rows, cols, nnz, W_all + m * B * nnz, B, N, N, in, N * C, C, o);
    if (rc) return rc;
    if (use_residual)
      for (int64_t i = 0; i < sz; ++i) o[i] = o[i] + V0[i]; /* V = V + res_conn */
  }
  return 0;
}

#undef FN
#undef CAT
#undef CAT_
