"""PSFNet module surface on top of the HIP chord-spmm kernels.

Four constructor signatures exist in the reference, all wrapping the same hot loop; each is mirrored by a
class here with the same argument names, attribute names and state_dict keys, so the reference's training /
inference scripts and its shipped checkpoints work unchanged:

=====================  =================================================  ==========================
class                  reference                                          drop-in module
=====================  =================================================  ==========================
``SyntheticPSFNet``    SyntheticExperiments/psf.py:62-191                  ``synthetic_psf.PSFNet``
``LRAPSFNet``          LRA/psf.py:63-250 (== LRA/attention_maps/psf.py)    ``lra_psf.PSFNet``
``GenomePSFNet``       Genome_Clf/psf.py:63-240                            ``genome_psf.PSFNet``
``AttentionBlockPSF``  attention_block.py:70-178                           ``attention_block.PSFNet``
=====================  =================================================  ==========================

State-dict layout kept: ``embedding.weight``, ``pos_embedding.weight`` (``apc_embedding`` in
the attention block), ``fs.{m}.network.{0,2}.{weight,bias}``, ``g.network.{0,2}.{weight,bias}``,
``final.*``, ``init_linear.*``. ``chord_indicies`` (sic) stays a plain attribute outside the state_dict.

What differs from the reference, on purpose: the M sparse products are not M calls into a generic
gather/multiply/scatter (torch_sparse.spmm) with a separate residual kernel each, but one call into
libpsf_chord.so that enqueues M HIP kernels with the residual fused (``chord_chain``). All W_m = fs[m](data)
depend only on ``data`` (psf.py:175), never on V, so they are produced first and the chain runs back to back.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Union

import torch
from torch import nn

from . import _lib, fused_mixer, fused_mlp
from .chord import chord_chain, chord_spmm, get_chord_indices_assym
from .token_linear import TokenEmbedding, TokenLinear, embed_tokens

LayerSpec = Sequence[Union[str, int]]


def MakeMLP(cfg: LayerSpec, in_channels: int, out_channels: int) -> nn.Sequential:
    """An int in ``cfg`` is a Linear to that width, any string is a GELU; a closing Linear maps to
    ``out_channels`` (SyntheticExperiments/psf.py:35-47). ``[32, 'GELU']`` gives Linear, GELU, Linear, i.e.
    parameter indices 0 and 2."""
    # TokenLinear is nn.Linear (same parameters / state_dict) whose weight and bias gradients over the ~1e6
    # tokens run on the tall-skinny MFMA kernel instead of a library GEMM (token_linear.py)
    stack: List[nn.Module] = []
    width = in_channels
    for item in cfg:
        if isinstance(item, int):
            stack.append(TokenLinear(width, item))
            width = item
        else:
            stack.append(nn.GELU())
    stack.append(TokenLinear(width, out_channels))
    return nn.Sequential(*stack)


class MLPBlock(nn.Module):
    """Token-wise MLP; the submodule is called ``network`` (SyntheticExperiments/psf.py:50-60)."""

    def __init__(self, cfg: LayerSpec, in_dim: int, out_dim: int):
        super().__init__()
        self.network = MakeMLP(cfg, in_dim, out_dim)

    def forward(self, data: torch.Tensor) -> torch.Tensor:
        return self.network(data)


class _FlatHeadFn(torch.autograd.Function):
    """out = flat @ weight.T + bias on ``psf_flat_head_f32``, its backward on ``psf_flat_head_bwd_f32`` (csrc/flat_head.hip)."""

    @staticmethod
    def forward(ctx, flat, weight, bias):
        lib = _lib.load()
        flat_c, w_c = flat.contiguous(), weight.contiguous()
        B, K = flat_c.shape
        J = w_c.shape[0]
        ctx.save_for_backward(flat_c, w_c)
        ctx.has_bias = bias is not None
        if J > 8:
            # The forward kernel holds J weight chunks in registers (J <= 8). CIFAR-10's first head layer (16 outputs, K = 16384)
            # keeps the library GEMM forward — in two groups of eight rows the kernel's 16 workgroups took 2 x 26 us against the
            # GEMM's 28 (profiles/r03ap_family_step_kernels.log) — and takes the streaming kernel backward (13 us).
            return torch.nn.functional.linear(flat_c, w_c, bias)
        ws_bytes = lib.psf_flat_head_workspace(B, K, J)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=flat.device)
        out = torch.empty((B, J), dtype=torch.float32, device=flat.device)
        with torch.cuda.device(flat.device):
            rc = lib.psf_flat_head_f32(flat_c.data_ptr(), w_c.data_ptr(), bias.data_ptr() if bias is not None else None,
                                       out.data_ptr(), B, K, J, ws.data_ptr(), ws_bytes,
                                       _lib.stream_ptr(flat.device))
        _lib.check(rc, "psf_flat_head_f32")
        return out

    @staticmethod
    def backward(ctx, dy):
        flat, weight = ctx.saved_tensors
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_flat = d_w = None
        B, K = flat.shape
        J = weight.shape[0]
        if (need_x or need_w) and dy.dtype == torch.float32 and B <= 1024 and flat.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0:
            # psf_flat_head_bwd_f32: one pass over X instead of two library GEMMs with K in the hundreds of thousands and
            # J, B tiny (272 us for dW at the genome shape; csrc/flat_head.hip)
            dy_c = dy.contiguous()
            d_flat = torch.empty_like(flat) if need_x else None
            d_w = torch.empty_like(weight) if need_w else None
            with torch.cuda.device(flat.device):
                rc = _lib.load().psf_flat_head_bwd_f32(dy_c.data_ptr(), flat.data_ptr(), weight.data_ptr(),
                                                       d_flat.data_ptr() if need_x else None, d_w.data_ptr() if need_w else None,
                                                       B, K, J, _lib.stream_ptr(flat.device))
            _lib.check(rc, "psf_flat_head_bwd_f32")
        else:
            d_flat = dy.mm(weight) if need_x else None
            d_w = dy.t().mm(flat) if need_w else None
        d_b = dy.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return d_flat, d_w, d_b


def _flat_head(final: nn.Module, flat: torch.Tensor) -> torch.Tensor:
    """``final(flat)`` for the FLATTEN head, Linear(N*C -> n_class) on [B, N*C] (psf.py:129-134). As a GEMM the
    library takes 0.57 ms (Adding, one output) or 117 us (Temporal Order, four) at N*C = 131072, B = 64
    (profiles/r01_e2e_forward_split.log); ``psf_flat_head_f32`` reads the activations once."""
    if isinstance(final, nn.Sequential) and len(final) > 0 and isinstance(final[0], nn.Linear):
        # the non-linear head (LRA/psf.py: Linear(N*C -> hidden), GELU, Linear(hidden -> n_class)): its first layer is the wide one
        return final[1:](_flat_head(final[0], flat))
    if (isinstance(final, nn.Linear) and final.out_features <= 16 and flat.is_cuda and flat.dim() == 2
            and flat.dtype == torch.float32 and final.weight.dtype == torch.float32 and flat.shape[1] % 4 == 0
            and flat.shape[1] >= 4096
            # psf_flat_head_f32 reads 16-byte vectors and returns PSF_E_ALIGN otherwise: a view at an odd storage offset (or
            # a weight view) takes the library GEMM instead of raising
            and flat.is_contiguous() and flat.data_ptr() % 16 == 0 and final.weight.is_contiguous()
            and final.weight.data_ptr() % 16 == 0):
        return _FlatHeadFn.apply(flat, final.weight, final.bias)
    return final(flat)


class _ChordMixer(nn.Module):
    """Shared machinery: the f/g networks and the chain V <- W_m V (+ V_0)."""

    #: run all M products in one library call (default) instead of one autograd node per factor
    fused_chain: bool = True

    def _build_mixer(self, seq_len: int, n_W: int, Ws: LayerSpec, V: LayerSpec, width_in: int, channels: int,
                     use_cuda: bool) -> None:
        self.fs = nn.ModuleList([MLPBlock(Ws, width_in, n_W + 1) for _ in range(n_W)])
        self.g = MLPBlock(V, width_in, channels)
        self._seq_len = seq_len

    def _build_indices(self, seq_len: int, n_links: int, use_cuda: bool) -> None:
        # Kept for scripts that pass it to spmm themselves (e.g. ChangedPSF, pathfinder_inference.py:66-81).
        # A plain attribute like in the reference (psf.py:143-145): not a parameter, not in the state_dict.
        idx = torch.tensor(get_chord_indices_assym(seq_len, n_links), dtype=torch.int64)
        self.chord_indicies = idx.cuda() if use_cuda else idx

    def _apply(self, fn, *args, **kwargs):
        # net.cuda() / net.to(device) also moves the index attribute (the reference leaves it behind).
        super()._apply(fn, *args, **kwargs)
        idx = getattr(self, "chord_indicies", None)
        if isinstance(idx, torch.Tensor):
            moved = fn(idx)
            if moved.dtype == torch.int64:
                self.chord_indicies = moved
        return self

    def link_weights(self, data: torch.Tensor) -> List[torch.Tensor]:
        """W_m = fs[m](data), each [B, N, L] (psf.py:175)."""
        fs = list(self.fs)
        if fused_mlp.eligible(data, fs):
            return fused_mlp.fused_mlp_forward(data, fs)
        if fused_mlp.trainable(data, fs):
            return fused_mlp.fused_mlp_apply(data, fs)
        if fused_mlp.wide_ok(data, fs):
            return fused_mlp.wide_apply(data, fs)
        if fused_mlp.stackable(data, fs):
            return fused_mlp.stacked_apply(data, fs)
        return [f(data) for f in fs]

    def produce(self, data: torch.Tensor):
        """(V, [W_m]) = (g(data), [fs[m](data)]) — psf.py:165,175. All M+1 MLPs run as ONE fused launch from one
        read of ``data``, and under autograd their backward is one fused launch too (fused_mlp.py); shapes the
        kernels do not cover use the PyTorch layers.
        Returns links = None when the chain is not fused (each W_m is then produced right before its step)."""
        if not self.fused_chain:
            return self.g(data), None
        blocks = [self.g] + list(self.fs)
        if fused_mlp.eligible(data, blocks):
            outs = fused_mlp.fused_mlp_forward(data, blocks)
            return outs[0], outs[1:]
        if fused_mlp.trainable(data, blocks):
            outs = fused_mlp.fused_mlp_apply(data, blocks)
            return outs[0], outs[1:]
        if fused_mlp.wide_ok(data, blocks):
            outs = fused_mlp.wide_apply(data, blocks)
            return outs[0], outs[1:]
        if fused_mlp.stackable(data, blocks):
            outs = fused_mlp.stacked_apply(data, blocks)
            return outs[0], outs[1:]
        return self.g(data), self.link_weights(data)

    def mix_from_recipe(self, recipe, use_residuals: bool):
        """V_M from the RECIPE of ``data`` (fused_mixer.Recipe: the affine input layer or the embedding lookup that PSFNet
        applies first, psf.py:151-162): neither ``data`` nor any W_m is written to memory. None when that path does not apply."""
        if recipe is not None and self.fused_chain and fused_mixer.eligible_recipe(recipe, self.g, list(self.fs)):
            return fused_mixer.mixer_forward_in(recipe, self.g, list(self.fs), use_residuals)
        return None

    def mix_from_data(self, data: torch.Tensor, use_residuals: bool):
        """V_M straight from ``data`` with every W_m computed inside its chain step and never written (fused_mixer.py,
        csrc/fwd_mlp_step.h) — psf.py:165-188 in M + 2 launches. None when that path does not apply (a gradient is
        needed, shapes outside its limits): the caller then runs ``produce`` + ``mix``."""
        if self.fused_chain and fused_mixer.eligible(data, self.g, list(self.fs)):
            return fused_mixer.mixer_forward(data, self.g, list(self.fs), use_residuals)
        return None

    def mix(self, data: torch.Tensor, V: torch.Tensor, use_residuals: bool, links=None) -> torch.Tensor:
        """The hot loop of PSFNet.forward (SyntheticExperiments/psf.py:167-188). ``links`` may carry
        precomputed W_m (from ``link_weights``)."""
        if self.fused_chain or links is not None:
            return chord_chain(self.link_weights(data) if links is None else links, V, use_residuals)
        res_conn = V if use_residuals else None
        for f in self.fs:
            V = chord_spmm(f(data), V, res_conn)
        return V


class SyntheticPSFNet(_ChordMixer):
    """Adding / Temporal-Order model — SyntheticExperiments/psf.py:62-191."""

    def __init__(self, vocab_size, add_init_linear_layer, embedding_size, n_vec, n_W, Ws, V, n_channels_V,
                 n_class, pooling_type, head, use_cuda, use_residuals, use_pos_embedding, problem):
        super().__init__()
        self.vocab_size = vocab_size
        self.add_init_linear_layer = add_init_linear_layer
        self.embedding_size = embedding_size
        self.n_vec = n_vec
        self.n_W = n_W
        self.n_links = n_W + 1
        self.Ws = Ws
        self.V = V
        self.n_channels_V = n_channels_V
        self.n_class = n_class
        self.pooling_type = pooling_type
        self.head = head
        self.use_cuda = use_cuda
        self.use_residuals = use_residuals
        self.use_pos_embedding = use_pos_embedding
        self.problem = problem

        # construction order follows the reference so a given torch seed draws the same initial weights
        self.embedding = TokenEmbedding(vocab_size, embedding_size)
        self.pos_embedding = nn.Embedding(n_vec, embedding_size)
        self._build_mixer(n_vec, n_W, Ws, V, embedding_size, n_channels_V, use_cuda)
        if head[0] == 'linear':
            self.final = nn.Linear(n_vec * n_channels_V, n_class, bias=True)
        if add_init_linear_layer:
            self.init_linear = TokenLinear(2, embedding_size, bias=True)
        self._build_indices(n_vec, self.n_links, use_cuda)

    def _recipe(self, data):
        """How ``forward`` obtains its ``data`` from the raw batch, when that is one lookup or one affine layer."""
        pos = self.pos_embedding.weight if self.use_pos_embedding else None
        if self.problem == 'order':
            if data.dim() == 3 and data.size(-1) == 1 and not self.add_init_linear_layer and data.dtype == torch.int64:
                return fused_mixer.Recipe.tokens(data.squeeze(-1), self.embedding.weight, pos)
            return None
        if self.add_init_linear_layer and data.dim() == 3 and data.is_floating_point() and data.size(-1) <= 3:
            return fused_mixer.Recipe.affine(data, self.init_linear, pos)
        return None

    def forward(self, data):
        V = self.mix_from_recipe(self._recipe(data), self.use_residuals)
        if V is not None:
            return _flat_head(self.final, V.reshape(V.size(0), -1))
        pos_done = False
        if self.problem == 'order':
            if data.dim() == 3 and data.size(-1) == 1:
                # lookup and positional add in one pass; pos_embedding(arange(n_vec)) per sample == its weight
                pos_done = self.use_pos_embedding and not self.add_init_linear_layer
                data = embed_tokens(data.squeeze(-1), self.embedding, self.pos_embedding.weight if pos_done else None)
            else:
                data = self.embedding(data).squeeze(-2)
        if self.add_init_linear_layer:
            data = self.init_linear(data)
        if self.use_pos_embedding and not pos_done:
            data = data + self.pos_embedding.weight.unsqueeze(0)  # == pos_embedding(arange(n_vec)) per sample
        V = self.mix_from_data(data, self.use_residuals)
        if V is None:
            V, links = self.produce(data)
            V = self.mix(data, V, self.use_residuals, links)
        return _flat_head(self.final, V.reshape(V.size(0), -1))


class _TokenPSFNet(_ChordMixer):
    """Common body of the LRA and Genome models (LRA/psf.py:63-250, Genome_Clf/psf.py:63-240)."""

    def _init_common(self, vocab_size, embedding_size, n_vec, n_W, Ws, V, n_channels_V, n_class, pooling_type,
                     head, use_cuda, use_residuals, dropout1_p, dropout2_p, dropout3_p, init_embedding_weights,
                     use_pos_embedding, padding_idx):
        self.vocab_size = vocab_size
        self.embedding_size = embedding_size
        self.n_vec = n_vec
        self.n_W = n_W
        self.n_links = n_W + 1
        self.Ws = Ws
        self.V = V
        self.n_channels_V = n_channels_V
        self.n_class = n_class
        self.pooling_type = pooling_type
        self.head = head
        self.use_cuda = use_cuda
        self.use_residuals = use_residuals
        self.dropout1_p = dropout1_p
        self.dropout2_p = dropout2_p
        self.dropout3_p = dropout3_p
        self.init_embedding_weights = init_embedding_weights
        self.use_pos_embedding = use_pos_embedding

        if padding_idx is None:
            self.embedding = TokenEmbedding(vocab_size, embedding_size)
        else:
            self.embedding = TokenEmbedding(vocab_size, embedding_size, padding_idx=padding_idx)
        self.pos_embedding = nn.Embedding(n_vec, embedding_size)
        if init_embedding_weights:
            self.init_embed_weights()
        self._build_mixer(n_vec, n_W, Ws, V, embedding_size, n_channels_V, use_cuda)

        pooled = n_vec * n_channels_V if pooling_type == 'FLATTEN' else n_channels_V
        if pooling_type in ('FLATTEN', 'CLS'):
            if head[0] == 'linear':
                self.final = nn.Linear(pooled, n_class)
            elif head[0] == 'non-linear':
                self.final = nn.Sequential(nn.Linear(pooled, head[1]), nn.GELU(), nn.Linear(head[1], n_class))

        self.dropout1 = nn.Dropout(dropout1_p)
        self.dropout2 = nn.Dropout(dropout2_p)
        self.dropout3 = nn.Dropout(dropout3_p)
        self._build_indices(n_vec, self.n_links, use_cuda)

    def init_embed_weights(self):
        """uniform(-0.1, 0.1) token embeddings (LRA/psf.py:192-195)."""
        self.embedding.weight.data.uniform_(-0.1, 0.1)
        self.embedding.weight.requires_grad = True

    def features(self, data, links=None):
        """Everything up to and including the chain and dropout3: returns V [B,N,C]. When ``links`` is a
        list it is filled with the W_m that were used."""
        quiet = not self.training or (self.dropout1.p == 0 and self.dropout2.p == 0)  # dropout1 / 2 sit inside what is fused
        if links is None and quiet and data.dim() == 2 and data.dtype == torch.int64:
            V = self.mix_from_recipe(fused_mixer.Recipe.tokens(data, self.embedding.weight,
                                                               self.pos_embedding.weight if self.use_pos_embedding else None),
                                     self.use_residuals)
            if V is not None:
                return self.dropout3(V)
        data = embed_tokens(data, self.embedding, self.pos_embedding.weight if self.use_pos_embedding else None)
        data = self.dropout1(data)
        if links is None and not (self.training and self.dropout2.p > 0):  # dropout2 sits between g and the loop
            V = self.mix_from_data(data, self.use_residuals)
            if V is not None:
                return self.dropout3(V)
        V, produced = self.produce(data)
        V = self.dropout2(V)
        if links is not None:  # the caller wants the W_m (attention-map extraction)
            if produced is None:
                produced = self.link_weights(data)
            links.extend(produced)
        V = self.mix(data, V, self.use_residuals, produced)
        return self.dropout3(V)

    def pool_and_classify(self, V):
        if self.pooling_type == 'CLS':
            V = V[:, 0, :]
        flat = V.reshape(V.size(0), -1)
        if self.pooling_type == 'FLATTEN':  # Linear(N*C -> n_class): the streaming head kernels where they apply
            return _flat_head(self.final, flat)
        return self.final(flat)

    def forward(self, data):
        return self.pool_and_classify(self.features(data))


class LRAPSFNet(_TokenPSFNet):
    """LRA model (ListOps / IMDb / CIFAR-10 / Pathfinder) — LRA/psf.py:63-250."""

    def __init__(self, vocab_size, embedding_size, n_vec, n_W, Ws, V, n_channels_V, n_class, pooling_type, head,
                 use_cuda, use_residuals, dropout1_p, dropout2_p, dropout3_p, init_embedding_weights,
                 use_pos_embedding, problem):
        super().__init__()
        self.problem = problem
        if problem in ('imdb', 'listops'):
            padding_idx = vocab_size - 2  # LRA/psf.py:106-111
        elif problem in ('cifar10', 'pathfinder'):
            padding_idx = None
        else:
            raise ValueError(f"unknown LRA problem '{problem}' (imdb, listops, cifar10, pathfinder)")
        self._init_common(vocab_size, embedding_size, n_vec, n_W, Ws, V, n_channels_V, n_class, pooling_type,
                          head, use_cuda, use_residuals, dropout1_p, dropout2_p, dropout3_p,
                          init_embedding_weights, use_pos_embedding, padding_idx)


class GenomePSFNet(_TokenPSFNet):
    """Genome classification model — Genome_Clf/psf.py:63-240 (the LRA model without ``problem``)."""

    def __init__(self, vocab_size, embedding_size, n_vec, n_W, Ws, V, n_channels_V, n_class, pooling_type, head,
                 use_cuda, use_residuals, dropout1_p, dropout2_p, dropout3_p, init_embedding_weights,
                 use_pos_embedding):
        super().__init__()
        self._init_common(vocab_size, embedding_size, n_vec, n_W, Ws, V, n_channels_V, n_class, pooling_type,
                          head, use_cuda, use_residuals, dropout1_p, dropout2_p, dropout3_p,
                          init_embedding_weights, use_pos_embedding, None)


class AttentionBlockPSF(_ChordMixer):
    """Stand-alone PSF attention block — attention_block.py:70-178. n_W = ceil(log2(max_seq_len)),
    C = embedding_size, returns the mixed sequence [B, N, E] instead of logits."""

    def __init__(self, vocab_size, embedding_size, max_seq_len, use_cuda, use_residuals, dropout1_p, dropout2_p,
                 dropout3_p):
        super().__init__()
        self.vocab_size = vocab_size
        self.embedding_size = embedding_size
        self.max_seq_len = max_seq_len
        self.n_W = math.ceil(math.log2(max_seq_len))
        self.n_links = self.n_W + 1
        self.Ws = [embedding_size, 'GELU']
        self.V = [embedding_size, 'GELU']
        self.use_cuda = use_cuda
        self.use_residuals = use_residuals
        self.dropout1_p = dropout1_p
        self.dropout2_p = dropout2_p
        self.dropout3_p = dropout3_p

        self._build_mixer(max_seq_len, self.n_W, self.Ws, self.V, embedding_size, embedding_size, use_cuda)
        self.dropout1 = nn.Dropout(dropout1_p)
        self.dropout2 = nn.Dropout(dropout2_p)
        self.dropout3 = nn.Dropout(dropout3_p)
        self._build_indices(max_seq_len, self.n_links, use_cuda)
        self.embedding = TokenEmbedding(vocab_size, embedding_size)
        self.apc_embedding = nn.Embedding(max_seq_len, embedding_size)

    def forward(self, data):
        if (not self.training or (self.dropout1.p == 0 and self.dropout2.p == 0)) and data.dim() == 2 and data.dtype == torch.int64:
            V = self.mix_from_recipe(fused_mixer.Recipe.tokens(data, self.embedding.weight, self.apc_embedding.weight),
                                     self.use_residuals)
            if V is not None:
                return self.dropout3(V)
        data = embed_tokens(data, self.embedding, self.apc_embedding.weight)
        data = self.dropout1(data)
        V = None if (self.training and self.dropout2.p > 0) else self.mix_from_data(data, self.use_residuals)
        if V is None:
            V, links = self.produce(data)
            V = self.mix(data, self.dropout2(V), self.use_residuals, links)
        return self.dropout3(V)


class ChangedPSF(LRAPSFNet):
    """LRAPSFNet that also returns the dense attention map ``W_M ... W_1`` — the ``ChangedPSF`` class of
    LRA/attention_maps/pathfinder_inference.py:30-94 and imdb_inference.py:24-71.

    The map is the same operator applied to ``eye(N)``: C = N channels, first operand unbatched
    (pathfinder_inference.py:57,75-81), no residual on the map.
    """

    def forward(self, data):
        Ws: List[torch.Tensor] = []
        V = self.features(data, Ws)  # dropout3 after the chain, as in the reference
        # The map is the chain applied to eye(N): C = N channels. The LDS-window kernels need C to be a multiple of four (16-byte
        # row chunks); IMDb's N = 4097 is not, and its map then ran on the generic kernel at 179 us per step (1.5 TB/s,
        # profiles/r03aq_family_infer_kernels.log). The identity is ours to shape: up to three zero columns are appended and
        # cut off again from the result (a view), which moves 0.07 % more bytes and keeps every product on the window kernels:
        # 133 us per step. (Padding on to whole 256-channel groups, 4352, does not shed the edge instance — W's size is not a
        # multiple of 16 bytes either — and costs its 6 %: 148 us.)
        n = self.n_vec
        cp = (n + 3) // 4 * 4
        if cp != n and V.is_cuda:
            eye = torch.zeros(n, cp, dtype=V.dtype, device=V.device)
            eye.diagonal().fill_(1)
            W_final = chord_chain(Ws, eye, False)[..., :n]
        else:
            eye = torch.eye(n, n, dtype=V.dtype, device=V.device)
            W_final = chord_chain(Ws, eye, False)
        return self.pool_and_classify(V), W_final


class GraphedInference:
    """``net(x)`` under ``torch.no_grad()`` captured once in a HIP graph and replayed — for the short-sequence models,
    whose forward is a few dozen launches for ~0.1 ms of GPU work (Adding N = 2048, B = 64: 0.27 ms eager, 0.13 ms
    replayed; at N = 16384 the forward is GPU-bound and a replay changes nothing —
    profiles/r01_e2e_forward_split.log). Fixed input shape; ``net`` in eval mode; the result tensors are static and
    overwritten by the next call."""

    def __init__(self, net: nn.Module, example: torch.Tensor):
        if not example.is_cuda:
            raise RuntimeError("GraphedInference needs a GPU tensor")
        self.net = net.eval()
        self.x = example.clone()
        with torch.no_grad():
            side = torch.cuda.Stream(device=example.device)
            side.wait_stream(torch.cuda.current_stream(example.device))
            with torch.cuda.stream(side):  # eager warm-up (allocator, lazy initialisation) off the capture stream
                self.net(self.x)
            torch.cuda.current_stream(example.device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = self.net(self.x)

    def __call__(self, x: torch.Tensor):
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.out
