"""Thin tensor-level wrappers over the C ABI: plans, workspace, and the three compute calls.

PyTorch is plumbing here (device memory, streams); all arithmetic happens in libreni_hip.so.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.RENILibraryError(
                "RENI HIP ops need tensors on a GPU device (got a CPU tensor); there is no CPU fallback")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class Plan:
    """Immutable kernel plan for one decoder architecture (reni_plan_create)."""

    def __init__(self, equivariance: str, ndims: int, hidden_features: int, hidden_layers: int,
                 out_features: int = 3, last_layer_linear: bool = True, output_activation=None,
                 first_omega_0: float = 30.0, hidden_omega_0: float = 30.0, dtype: str = "f32",
                 conditioning: str = "concat", mapping_layers: int = 0, mapping_features: int = 0):
        lib = _lib.load()
        if equivariance not in _lib.EQ:
            raise ValueError(f"equivariance {equivariance!r}")
        if output_activation not in _lib.ACT:
            raise ValueError(f"output_activation {output_activation!r}")
        self.desc = _lib.reni_desc(_lib.EQ[equivariance], ndims, hidden_features, hidden_layers, out_features,
                                   1 if last_layer_linear else 0, _lib.ACT[output_activation],
                                   float(first_omega_0), float(hidden_omega_0), _lib.DTYPE[dtype],
                                   {"concat": _lib.COND_CONCAT, "film": _lib.COND_FILM}[conditioning],
                                   int(mapping_layers), int(mapping_features))
        self.conditioning = conditioning
        self.hidden_features = hidden_features
        self.hidden_layers = hidden_layers
        self._h = ctypes.c_void_p()
        _lib.check(lib.reni_plan_create(ctypes.byref(self.desc), ctypes.byref(self._h)))
        self.lib = lib
        self.ndims = ndims
        self.dtype = dtype
        self.n_params = int(lib.reni_param_count(self._h))
        self.n_map_params = int(lib.reni_film_map_param_count(self._h))
        self.in_features = int(lib.reni_in_features(self._h))
        self.n_first = hidden_features * self.in_features + hidden_features   # W0 + b0: the head of the flat buffer
        self._ws = {}

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.reni_plan_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:
            pass

    # ---- workspace (a torch uint8 tensor, cached per device and size)
    def workspace(self, B: int, P: int, flags: int, device) -> torch.Tensor:
        n = int(self.lib.reni_workspace_bytes(self._h, B, P, flags))
        if n == 0:
            raise _lib.RENILibraryError("reni_workspace_bytes returned 0 (unsupported configuration)")
        key = (torch.device(device).index, )
        ws = self._ws.get(key)
        if ws is None or ws.numel() < n:
            ws = torch.empty(n + 256, dtype=torch.uint8, device=device)
            self._ws[key] = ws
        # Every hand-out counts: reni_train_step_rows keeps the NEXT batch's prologue (gathered rows, A_b, layer-0 operands, packed
        # weight images) in this buffer between two calls, and any other call on the plan -- a validation forward at another B, a
        # fused loss, a reallocation for a larger problem -- overwrites it at shifted offsets or abandons the buffer.  train_step
        # compares this counter with the one it saw at the end of the step that staged.
        self._ws_gen = getattr(self, "_ws_gen", 0) + 1
        return ws

    @staticmethod
    def _aligned_ptr(ws: torch.Tensor) -> Tuple[int, int]:
        p = ws.data_ptr()
        ap = (p + 255) & ~255
        return ap, ws.numel() - (ap - p)

    def launch_info(self, B: int, P: int):
        info = (ctypes.c_int32 * 4)()
        _lib.check(self.lib.reni_launch_info(self._h, B, P, info))
        return {"workgroups": info[0], "threads": info[1], "lds_bytes": info[2], "tiles": info[3]}

    def path_info(self, B: int, P: int, need_dw: bool = True, need_dz: bool = True):
        """Which kernels a backward call of this shape takes (reni_path_info): printed by bench.py so that an environment selector
        left set by accident shows on the line."""
        info = (ctypes.c_int32 * 8)()
        flags = (_lib.NEED_DW if need_dw else 0) | (_lib.NEED_DZ if need_dz else 0)
        _lib.check(self.lib.reni_path_info(self._h, B, P, flags, info))
        env = [n for b, n in ((1, "RENI_NO_PERSIST"), (2, "RENI_NO_SIDE_STREAM"), (4, "RENI_FRAG_WS_CAP_MB"), (8, "RENI_DW1_OLD"),
                              (16, "RENI_NO_L0X")) if info[6] & b]
        # (k_reni_l0_ring: layer 0's backward + dW_1 behind the L0X training instance -- given, in addition, WeightedMSE and no output image)
        return {"persistent_kernels": bool(info[0]), "dw1_kernel": ("none", "k_reni_dw1_ring", "k_reni_dw1", "k_reni_l0_ring")[info[1]],
                "side_stream": bool(info[2]), "images_per_chunk": info[3], "operand_stream": bool(info[4]),
                "fragment_stream": ("none", "bf16", "f32")[info[5]], "env_overrides": env, "workgroups": info[7]}

    # ---- compute
    def _grid_args(self, Z, D):
        B = Z.shape[0]
        if D.dim() == 2:
            D = D.unsqueeze(0)
        if D.shape[0] not in (1, B):
            raise ValueError(f"directions batch {D.shape[0]} does not match latent batch {B}")
        P = D.shape[1]
        dbs = 0 if D.shape[0] == 1 or D.stride(0) == 0 else P * 3
        if dbs == 0:
            D = D[:1]
        return B, P, _f32c(D), dbs

    def _check_zp(self, Z, params, map_params=None):
        """The C ABI takes bare pointers: a mis-shaped latent or parameter buffer would be read out of bounds."""
        if Z.dim() != 3 or tuple(Z.shape[1:]) != (self.ndims, 3):
            raise ValueError(f"Z must be [B,{self.ndims},3], got {tuple(Z.shape)}")
        if params.numel() != self.n_params:
            raise ValueError(f"params must hold {self.n_params} floats, got {params.numel()}")
        if map_params is not None and map_params.numel() != self.n_map_params:
            raise ValueError(f"map_params must hold {self.n_map_params} floats, got {map_params.numel()}")

    def forward(self, Z: torch.Tensor, D: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
        _require_cuda(Z, D, params)
        Z = _f32c(Z); params = _f32c(params)
        self._check_zp(Z, params)
        B, P, Dc, dbs = self._grid_args(Z, D)
        out = torch.empty(B, P, 3, dtype=torch.float32, device=Z.device)
        ws = self.workspace(B, P, 0, Z.device)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(Z.device).cuda_stream
        _lib.check(self.lib.reni_forward(self._h, B, P, Z.data_ptr(), Dc.data_ptr(), dbs, params.data_ptr(),
                                         out.data_ptr(), wp, wn, stream))
        return out

    def forward_loss_backward(self, Z, D, params, target, weight, loss_kind="mse", alpha=0.0, beta=0.0,
                              need_dw=True, need_dz=True, want_out=False, idx=None, sparse_weight=False):
        """target / weight: any strided [B,P,3] views (stride 0 broadcasts); returns
        (loss_terms[4] device tensor, dZ or None, dparams or None, out or None).
        idx (int64 device tensor [B]): Z is a latent TABLE and image b uses row idx[b] (gathered inside the prologue
        kernel: reni_forward_loss_backward_rows); dZ stays [B,ND,3] in batch order.  An index outside the table makes the
        call's loss and gradients NaN (no out-of-bounds read, no host synchronisation to check it).
        sparse_weight: the weight is zero over whole regions (an inpainting mask, RENI_module.py:92-94): RENI_WEIGHT_SPARSE --
        the frozen-decoder persistent kernels then leave out the tiles (and the statistics pass) that cannot change the result --
        bit-equal to the dense call.  sparse_weight="pixels": RENI_WEIGHT_COMPACT -- the pixels with weight are also packed into
        each image's first tiles (fewer tiles still; the sums are re-associated: equal to fp32 rounding)."""
        _require_cuda(Z, D, params, target, weight, idx)
        Z = _f32c(Z); params = _f32c(params)
        self._check_zp(Z, params)
        if idx is not None:
            if idx.dtype != torch.int64 or idx.dim() != 1 or idx.numel() < 1:
                raise ValueError("idx must be a non-empty 1-D int64 tensor")
            idx = idx.contiguous()
            B, P, Dc, dbs = self._grid_args(Z[:1].expand(idx.numel(), -1, -1), D)  # (the batch's shape; Z stays the table)
        else:
            B, P, Dc, dbs = self._grid_args(Z, D)
        if target.dtype != torch.float32:
            target = target.float()
        if weight.dtype != torch.float32:
            weight = weight.float()
        target = target.expand(B, P, 3)
        weight = weight.expand(B, P, 3)
        ts = (ctypes.c_int64 * 3)(*target.stride())
        wst = (ctypes.c_int64 * 3)(*weight.stride())
        if sparse_weight not in (False, True, None, "tiles", "pixels"):
            raise ValueError('sparse_weight must be False, True / "tiles", or "pixels"')
        flags = ((_lib.NEED_DW if need_dw else 0) | (_lib.NEED_DZ if need_dz else 0)
                 | (_lib.WEIGHT_COMPACT if sparse_weight == "pixels" else _lib.WEIGHT_SPARSE if sparse_weight else 0))
        dev = Z.device
        loss_terms = torch.empty(4, dtype=torch.float32, device=dev)
        dZ = torch.empty(B, self.ndims, 3, dtype=torch.float32, device=dev) if need_dz else None
        dparams = torch.empty(self.n_params, dtype=torch.float32, device=dev) if need_dw else None
        out = torch.empty(B, P, 3, dtype=torch.float32, device=dev) if want_out else None
        ws = self.workspace(B, P, flags, dev)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(dev).cuda_stream
        kind = {"mse": _lib.LOSS_MSE, "test": _lib.LOSS_TEST}[loss_kind]
        tail = (Dc.data_ptr(), dbs, params.data_ptr(), target.data_ptr(), ts, weight.data_ptr(), wst, kind, float(alpha),
                float(beta), flags, out.data_ptr() if out is not None else None, loss_terms.data_ptr(),
                dZ.data_ptr() if dZ is not None else None, dparams.data_ptr() if dparams is not None else None, wp, wn, stream)
        if idx is not None:
            _lib.check(self.lib.reni_forward_loss_backward_rows(self._h, B, P, Z.data_ptr(), Z.shape[0], idx.data_ptr(), *tail))
        else:
            _lib.check(self.lib.reni_forward_loss_backward(self._h, B, P, Z.data_ptr(), *tail))
        return loss_terms, dZ, dparams, out

    def train_step(self, Z_table, idx, D, params, target, weight, m_dec, v_dec, m_lat, v_lat, step, lr, stage_state, idx_next=None,
                   loss_kind="mse", alpha=0.0, beta=0.0, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0, comm=None, overlap=False):
        """reni_train_step_rows: fused fwd + loss + bwd on the rows `idx` of the latent TABLE, then torch.optim.Adam's update of
        `params` and of the whole table, IN PLACE -- the same results as forward_loss_backward(idx=...) + adam_step2, with Adam and
        the next batch's prologue (`idx_next`) scheduled beside the backward pass's last kernel.  stage_state: a ctypes.c_uint32
        the caller keeps between calls (zero it whenever params / the table / the shapes change behind the library's back).
        comm: a ``dist.RcclComm`` -- the data-parallel step (reni_train_step_rows_dp): the decoder gradient is summed over the
        communicator's ranks INSIDE the call (pass grad_scale = 1 / world); overlap: the early slice on the library's own stream.
        Returns (loss_terms[4], dZ [B,ND,3], dparams)."""
        _require_cuda(Z_table, D, params, target, weight, idx, idx_next, m_dec, v_dec, m_lat, v_lat)
        for t in (Z_table, params, m_dec, v_dec, m_lat, v_lat):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("train_step updates its buffers in place: contiguous float32 tensors only")
        self._check_zp(Z_table, params)
        for ix in (idx, idx_next):
            if ix is not None and (ix.dtype != torch.int64 or ix.dim() != 1 or ix.numel() != idx.numel() or not ix.is_contiguous()):
                raise ValueError("idx / idx_next must be contiguous 1-D int64 tensors of the same length")
        B, P, Dc, dbs = self._grid_args(Z_table[:1].expand(idx.numel(), -1, -1), D)
        target = (target if target.dtype == torch.float32 else target.float()).expand(B, P, 3)
        weight = (weight if weight.dtype == torch.float32 else weight.float()).expand(B, P, 3)
        ts = (ctypes.c_int64 * 3)(*target.stride())
        wst = (ctypes.c_int64 * 3)(*weight.stride())
        dev = Z_table.device
        loss_terms = torch.empty(4, dtype=torch.float32, device=dev)
        dZ = torch.empty(B, self.ndims, 3, dtype=torch.float32, device=dev)
        dparams = torch.empty(self.n_params, dtype=torch.float32, device=dev)
        # a staged prologue is only good if NOTHING took the plan's workspace since the call that staged it (see workspace())
        if stage_state.value & 1 and getattr(self, "_stage_gen", None) != (getattr(self, "_ws_gen", 0), dev.index):
            stage_state.value = 0
        ws = self.workspace(B, P, _lib.NEED_DW | _lib.NEED_DZ, dev)
        self._stage_gen = (self._ws_gen, dev.index)
        wp, wn = self._aligned_ptr(ws)
        kind = {"mse": _lib.LOSS_MSE, "test": _lib.LOSS_TEST}[loss_kind]
        head = (self._h, B, P, Z_table.data_ptr(), Z_table.shape[0], idx.data_ptr(), idx_next.data_ptr() if idx_next is not None else None,
                Dc.data_ptr(), dbs, params.data_ptr(), target.data_ptr(), ts, weight.data_ptr(), wst, kind, float(alpha), float(beta),
                m_dec.data_ptr(), v_dec.data_ptr(), m_lat.data_ptr(), v_lat.data_ptr(), float(lr), float(betas[0]), float(betas[1]),
                float(eps), int(step), float(grad_scale))
        tail = (ctypes.byref(stage_state), loss_terms.data_ptr(), dZ.data_ptr(), dparams.data_ptr(), wp, wn,
                torch.cuda.current_stream(dev).cuda_stream)
        with torch.cuda.device(dev):
            if comm is not None:
                _lib.check(self.lib.reni_train_step_rows_dp(*head, comm._comm, 1 if overlap else 0, *tail))
            else:
                _lib.check(self.lib.reni_train_step_rows(*head, *tail))
        return loss_terms, dZ, dparams

    def weight_lists(self, B, P, weight, mode):
        """The tile / pixel lists of a masked weight (reni_weight_lists_build), built ONCE per (weight tensor, its version counter,
        shape, strides, mode) and kept on the plan: the inpainting mask of a FIT_LATENT run does not change over its epochs.
        weight: the [B, P, 3] (expanded) view the step passes; mode: _lib.WEIGHT_SPARSE or _lib.WEIGHT_COMPACT."""
        # CAVEAT: the key sees in-place writes only through torch's version counter -- a write through `weight.data`, a
        # custom kernel or ctypes does not bump it.  Whoever changes the mask that way calls invalidate_weight_lists() (or passes
        # cache_lists=False to latent_step).  Inference tensors have no version counter: they are never cached (-> None).
        try:
            version = weight._version
        except RuntimeError:
            return None
        key = (weight.data_ptr(), version, tuple(weight.shape), tuple(weight.stride()), int(B), int(P), int(mode), weight.device.index)
        cache = self.__dict__.setdefault("_wlists", {})
        hit = cache.get("key") == key
        if not hit:
            n = int(self.lib.reni_weight_lists_bytes(B, P))
            buf = cache.get("buf")
            if buf is None or buf.numel() < n + 256 or buf.device != weight.device:
                buf = torch.empty(n + 256, dtype=torch.uint8, device=weight.device)
            lp = (buf.data_ptr() + 255) & ~255
            wst = (ctypes.c_int64 * 3)(*weight.stride())
            summary = (ctypes.c_int32 * 3)()
            with torch.cuda.device(weight.device):
                _lib.check(self.lib.reni_weight_lists_build(B, P, weight.data_ptr(), wst, int(mode), lp, buf.numel() - (lp - buf.data_ptr()),
                                                            summary, torch.cuda.current_stream(weight.device).cuda_stream))
            # (one synchronisation per mask: no image with a live cosine term -> the steps leave the statistics launches out)
            cache.update(key=key, buf=buf, ptr=lp, weight=weight, cos_constant=summary[2] == 0,   # (the weight is kept alive: its
                         summary=tuple(summary))                                                  #  data_ptr is part of the key)
        return cache["ptr"], (_lib.WEIGHT_COS_CONSTANT if cache["cos_constant"] else 0)

    def invalidate_weight_lists(self):
        """Forget the cached tile / pixel lists: the next latent_step(sparse_weight=...) rebuilds them from the weight it is given.  To be
        called after a write into the weight tensor that torch's version counter cannot see (``weight.data``, a custom kernel, ctypes)."""
        self.__dict__.pop("_wlists", None)

    def latent_step(self, Z_table, idx, D, params, target, weight, m_lat, v_lat, step, lr, loss_kind="test", alpha=0.0, beta=0.0,
                    betas=(0.9, 0.999), eps=1e-8, sparse_weight=False, cache_lists=True):
        """reni_latent_step_rows: one FIT_LATENT iteration (frozen decoder) -- forward_loss_backward(idx=..., need_dw=False) then
        adam_rows_step on the table, IN PLACE, as one library call.  Returns (loss_terms[4], dZ [B,ND,3])."""
        _require_cuda(Z_table, D, params, target, weight, idx, m_lat, v_lat)
        for t in (Z_table, m_lat, v_lat):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("latent_step updates its buffers in place: contiguous float32 tensors only")
        params = _f32c(params)
        self._check_zp(Z_table, params)
        if idx.dtype != torch.int64 or idx.dim() != 1 or idx.numel() < 1 or not idx.is_contiguous():
            raise ValueError("idx must be a non-empty contiguous 1-D int64 tensor")
        if sparse_weight not in (False, True, None, "tiles", "pixels"):
            raise ValueError('sparse_weight must be False, True / "tiles", or "pixels"')
        B, P, Dc, dbs = self._grid_args(Z_table[:1].expand(idx.numel(), -1, -1), D)
        target = (target if target.dtype == torch.float32 else target.float()).expand(B, P, 3)
        weight = (weight if weight.dtype == torch.float32 else weight.float()).expand(B, P, 3)
        ts = (ctypes.c_int64 * 3)(*target.stride())
        wst = (ctypes.c_int64 * 3)(*weight.stride())
        dev = Z_table.device
        loss_terms = torch.empty(4, dtype=torch.float32, device=dev)
        dZ = torch.empty(B, self.ndims, 3, dtype=torch.float32, device=dev)
        flags = _lib.WEIGHT_COMPACT if sparse_weight == "pixels" else _lib.WEIGHT_SPARSE if sparse_weight else 0
        ws = self.workspace(B, P, _lib.NEED_DZ | flags, dev)
        wp, wn = self._aligned_ptr(ws)
        kind = {"mse": _lib.LOSS_MSE, "test": _lib.LOSS_TEST}[loss_kind]
        head = (self._h, B, P, Z_table.data_ptr(), Z_table.shape[0], idx.data_ptr(), Dc.data_ptr(), dbs, params.data_ptr(),
                target.data_ptr(), ts, weight.data_ptr(), wst, kind, float(alpha), float(beta), flags)
        tail = (m_lat.data_ptr(), v_lat.data_ptr(), float(lr), float(betas[0]), float(betas[1]), float(eps), int(step),
                loss_terms.data_ptr(), dZ.data_ptr(), wp, wn, torch.cuda.current_stream(dev).cuda_stream)
        chunked = False
        if flags:  # (H = 256 problems that run in image chunks keep the rebuilding entry point; decided once per shape)
            cc = self.__dict__.setdefault("_chunk_cache", {})
            if (B, P) not in cc:
                cc[(B, P)] = self.path_info(B, P, need_dw=False)["images_per_chunk"] < B
            chunked = cc[(B, P)]
        if flags and cache_lists and not chunked:
            # the lists from the plan's cache (built once per mask: reni_weight_lists_build), the step without its list-building launches
            wl = self.weight_lists(B, P, weight, flags)
        else:
            wl = None
        if wl is not None:
            lptr, extra = wl
            _lib.check(self.lib.reni_latent_step_rows_cached(*head[:-1], flags | extra, lptr, *tail))
        else:
            _lib.check(self.lib.reni_latent_step_rows(*head, *tail))
        return loss_terms, dZ

    def backward(self, Z, D, params, dout, need_dw=True, need_dz=True):
        _require_cuda(Z, D, params, dout)
        Z = _f32c(Z); params = _f32c(params); dout = _f32c(dout)
        self._check_zp(Z, params)
        B, P, Dc, dbs = self._grid_args(Z, D)
        if tuple(dout.shape) != (B, P, 3):
            raise ValueError(f"dout must be [{B},{P},3], got {tuple(dout.shape)}")
        flags = (_lib.NEED_DW if need_dw else 0) | (_lib.NEED_DZ if need_dz else 0)
        dev = Z.device
        dZ = torch.empty_like(Z) if need_dz else None
        dparams = torch.empty(self.n_params, dtype=torch.float32, device=dev) if need_dw else None
        ws = self.workspace(B, P, flags, dev)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self.lib.reni_backward(
            self._h, B, P, Z.data_ptr(), Dc.data_ptr(), dbs, params.data_ptr(), dout.data_ptr(), flags,
            dZ.data_ptr() if dZ is not None else None, dparams.data_ptr() if dparams is not None else None,
            wp, wn, stream))
        return dZ, dparams


    # ---- FiLM conditioning: the per-sample core (per-image glue in reni_amd/film.py)
    def _film_args(self, A, film, D):
        B = A.shape[0]
        H, L = self.hidden_features, self.hidden_layers
        assert A.shape == (B, H, 8), f"A must be [B,{H},8], got {tuple(A.shape)}"
        assert film.shape == (B, L, 2, H), f"film must be [B,{L},2,{H}], got {tuple(film.shape)}"
        if D.dim() == 2:
            D = D.unsqueeze(0)
        if D.shape[0] not in (1, B):
            raise ValueError(f"directions batch {D.shape[0]} does not match latent batch {B}")
        P = D.shape[1]
        dbs = 0 if D.shape[0] == 1 or D.stride(0) == 0 else P * 3
        if dbs == 0:
            D = D[:1]
        return B, P, _f32c(D), dbs

    def film_forward(self, A, film, D, params):
        _require_cuda(A, film, D, params)
        A = _f32c(A); film = _f32c(film); params = _f32c(params)
        B, P, Dc, dbs = self._film_args(A, film, D)
        assert params.numel() == self.n_params
        out = torch.empty(B, P, 3, dtype=torch.float32, device=A.device)
        ws = self.workspace(B, P, 0, A.device)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(A.device).cuda_stream
        _lib.check(self.lib.reni_film_forward(self._h, B, P, Dc.data_ptr(), dbs, A.data_ptr(), film.data_ptr(),
                                              params.data_ptr(), out.data_ptr(), wp, wn, stream))
        return out

    def film_forward_loss_backward(self, A, film, D, params, target, weight, loss_kind="mse", beta=0.0,
                                   need_dw=True, want_out=False):
        """-> (loss_terms[4], dA [B,H,8], dfilm [B,L,2,H], dparams or None, out or None)"""
        _require_cuda(A, film, D, params, target, weight)
        A = _f32c(A); film = _f32c(film); params = _f32c(params)
        B, P, Dc, dbs = self._film_args(A, film, D)
        if target.dtype != torch.float32:
            target = target.float()
        if weight.dtype != torch.float32:
            weight = weight.float()
        target = target.expand(B, P, 3)
        weight = weight.expand(B, P, 3)
        ts = (ctypes.c_int64 * 3)(*target.stride())
        wst = (ctypes.c_int64 * 3)(*weight.stride())
        flags = _lib.NEED_DW if need_dw else 0
        dev = A.device
        loss_terms = torch.empty(4, dtype=torch.float32, device=dev)
        dA = torch.empty_like(A)
        dfilm = torch.empty_like(film)
        dparams = torch.empty(self.n_params, dtype=torch.float32, device=dev) if need_dw else None
        out = torch.empty(B, P, 3, dtype=torch.float32, device=dev) if want_out else None
        ws = self.workspace(B, P, flags, dev)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(dev).cuda_stream
        kind = {"mse": _lib.LOSS_MSE, "test": _lib.LOSS_TEST}[loss_kind]
        _lib.check(self.lib.reni_film_forward_loss_backward(
            self._h, B, P, Dc.data_ptr(), dbs, A.data_ptr(), film.data_ptr(), params.data_ptr(), target.data_ptr(), ts,
            weight.data_ptr(), wst, kind, float(beta), flags, out.data_ptr() if out is not None else None,
            loss_terms.data_ptr(), dA.data_ptr(), dfilm.data_ptr(),
            dparams.data_ptr() if dparams is not None else None, wp, wn, stream))
        return loss_terms, dA, dfilm, dparams, out

    def film_backward(self, A, film, D, params, dout, need_dw=True):
        _require_cuda(A, film, D, params, dout)
        A = _f32c(A); film = _f32c(film); params = _f32c(params); dout = _f32c(dout)
        B, P, Dc, dbs = self._film_args(A, film, D)
        flags = _lib.NEED_DW if need_dw else 0
        dev = A.device
        dA = torch.empty_like(A)
        dfilm = torch.empty_like(film)
        dparams = torch.empty(self.n_params, dtype=torch.float32, device=dev) if need_dw else None
        ws = self.workspace(B, P, flags, dev)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self.lib.reni_film_backward(
            self._h, B, P, Dc.data_ptr(), dbs, A.data_ptr(), film.data_ptr(), params.data_ptr(), dout.data_ptr(), flags,
            dA.data_ptr(), dfilm.data_ptr(), dparams.data_ptr() if dparams is not None else None, wp, wn, stream))
        return dA, dfilm, dparams


    # ---- FiLM, whole model (per-image glue in HIP too): latents + mapping-network parameters in, their gradients out
    def film_model_forward(self, Z, D, params, map_params):
        _require_cuda(Z, D, params, map_params)
        Z = _f32c(Z); params = _f32c(params); map_params = _f32c(map_params)
        self._check_zp(Z, params, map_params)
        B, P, Dc, dbs = self._grid_args(Z, D)
        out = torch.empty(B, P, 3, dtype=torch.float32, device=Z.device)
        ws = self.workspace(B, P, 0, Z.device)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(Z.device).cuda_stream
        _lib.check(self.lib.reni_film_model_forward(self._h, B, P, Z.data_ptr(), Dc.data_ptr(), dbs, params.data_ptr(),
                                                    map_params.data_ptr(), out.data_ptr(), wp, wn, stream))
        return out

    def film_model_forward_loss_backward(self, Z, D, params, map_params, target, weight, loss_kind="mse", alpha=0.0,
                                         beta=0.0, need_dw=True, want_out=False):
        """-> (loss_terms[4], dZ, dparams or None, dmap_params or None, out or None).  The two gradients are the two halves
        of ONE buffer (dparams first), so an optimiser over [params | map_params] stored the same way takes it whole
        (``dparams._base``)."""
        _require_cuda(Z, D, params, map_params, target, weight)
        Z = _f32c(Z); params = _f32c(params); map_params = _f32c(map_params)
        self._check_zp(Z, params, map_params)
        B, P, Dc, dbs = self._grid_args(Z, D)
        if target.dtype != torch.float32:
            target = target.float()
        if weight.dtype != torch.float32:
            weight = weight.float()
        target = target.expand(B, P, 3)
        weight = weight.expand(B, P, 3)
        ts = (ctypes.c_int64 * 3)(*target.stride())
        wst = (ctypes.c_int64 * 3)(*weight.stride())
        flags = (_lib.NEED_DW if need_dw else 0) | _lib.NEED_DZ
        dev = Z.device
        loss_terms = torch.empty(4, dtype=torch.float32, device=dev)
        dZ = torch.empty_like(Z)
        gall = torch.empty(self.n_params + self.n_map_params, dtype=torch.float32, device=dev) if need_dw else None
        dparams = gall[:self.n_params] if need_dw else None
        dmap = gall[self.n_params:] if need_dw else None
        out = torch.empty(B, P, 3, dtype=torch.float32, device=dev) if want_out else None
        ws = self.workspace(B, P, flags, dev)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(dev).cuda_stream
        kind = {"mse": _lib.LOSS_MSE, "test": _lib.LOSS_TEST}[loss_kind]
        _lib.check(self.lib.reni_film_model_forward_loss_backward(
            self._h, B, P, Z.data_ptr(), Dc.data_ptr(), dbs, params.data_ptr(), map_params.data_ptr(), target.data_ptr(), ts,
            weight.data_ptr(), wst, kind, float(alpha), float(beta), flags, out.data_ptr() if out is not None else None,
            loss_terms.data_ptr(), dZ.data_ptr(), dparams.data_ptr() if need_dw else None,
            dmap.data_ptr() if need_dw else None, wp, wn, stream))
        return loss_terms, dZ, dparams, dmap, out

    def film_model_backward(self, Z, D, params, map_params, dout, need_dw=True):
        _require_cuda(Z, D, params, map_params, dout)
        Z = _f32c(Z); params = _f32c(params); map_params = _f32c(map_params); dout = _f32c(dout)
        self._check_zp(Z, params, map_params)
        B, P, Dc, dbs = self._grid_args(Z, D)
        if tuple(dout.shape) != (B, P, 3):
            raise ValueError(f"dout must be [{B},{P},3], got {tuple(dout.shape)}")
        flags = (_lib.NEED_DW if need_dw else 0) | _lib.NEED_DZ
        dev = Z.device
        dZ = torch.empty_like(Z)
        dparams = torch.empty(self.n_params, dtype=torch.float32, device=dev) if need_dw else None
        dmap = torch.empty(self.n_map_params, dtype=torch.float32, device=dev) if need_dw else None
        ws = self.workspace(B, P, flags, dev)
        wp, wn = self._aligned_ptr(ws)
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(self.lib.reni_film_model_backward(
            self._h, B, P, Z.data_ptr(), Dc.data_ptr(), dbs, params.data_ptr(), map_params.data_ptr(), dout.data_ptr(), flags,
            dZ.data_ptr(), dparams.data_ptr() if need_dw else None, dmap.data_ptr() if need_dw else None, wp, wn, stream))
        return dZ, dparams, dmap


def adam_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int, lr: float,
              betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
    """In-place torch.optim.Adam step on flat fp32 buffers (reni_adam_step)."""
    _require_cuda(p, g, m, v)
    lib = _lib.load()
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    stream = torch.cuda.current_stream(p.device).cuda_stream
    _lib.check(lib.reni_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr),
                                  float(betas[0]), float(betas[1]), float(eps), int(step), float(grad_scale), stream))


def adam_rows_step(table: torch.Tensor, g_rows: torch.Tensor, idx: torch.Tensor, m: torch.Tensor, v: torch.Tensor, step: int,
                   lr: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
    """In-place dense Adam step on table [N, ...] with the gradient g_rows [B, ...] of rows idx (reni_adam_rows_step)."""
    _require_cuda(table, g_rows, idx, m, v)
    lib = _lib.load()
    assert table.is_contiguous() and g_rows.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    idx = idx.to(torch.int64).contiguous()
    n_rows = table.shape[0]
    row_len = table.numel() // max(n_rows, 1)
    assert g_rows.numel() == idx.numel() * row_len
    stream = torch.cuda.current_stream(table.device).cuda_stream
    _lib.check(lib.reni_adam_rows_step(table.data_ptr(), g_rows.data_ptr(), idx.data_ptr(), idx.numel(), row_len, m.data_ptr(),
                                       v.data_ptr(), n_rows, float(lr), float(betas[0]), float(betas[1]), float(eps), int(step),
                                       float(grad_scale), stream))


def adam_step2(p, g, m, v, table, g_rows, idx, tm, tv, step: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8,
               grad_scale: float = 1.0):
    """adam_step on the flat decoder buffer and adam_rows_step on the latent table in ONE launch (reni_adam_step2)."""
    _require_cuda(p, g, m, v, table, g_rows, idx, tm, tv)
    lib = _lib.load()
    for t in (p, g, m, v, table, g_rows, tm, tv):
        assert t.is_contiguous()
    idx = idx.to(torch.int64).contiguous()
    n_rows = table.shape[0]
    row_len = table.numel() // max(n_rows, 1)
    assert g_rows.numel() == idx.numel() * row_len
    stream = torch.cuda.current_stream(p.device).cuda_stream
    with torch.cuda.device(p.device):
        _lib.check(lib.reni_adam_step2(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), table.data_ptr(),
                                       g_rows.data_ptr(), idx.data_ptr(), idx.numel(), row_len, tm.data_ptr(), tv.data_ptr(),
                                       n_rows, float(lr), float(betas[0]), float(betas[1]), float(eps), int(step),
                                       float(grad_scale), stream))


def selftest_layouts():
    lib = _lib.load()
    out = (ctypes.c_int32 * 2)()
    _lib.check(lib.reni_selftest_layouts(out, 2))
    return list(out)


def profile_enable(on: bool = True):
    """Record HIP events around every fused forward+backward kernel launch (reni_profile_enable)."""
    _lib.check(_lib.load().reni_profile_enable(1 if on else 0))


PROF_FWD_BWD, PROF_STATS, PROF_FWD, PROF_DW1, PROF_DWS, PROF_COMM, PROF_ALL = 0, 1, 2, 3, 4, 5, -1


def profile_minmax(kind: int = PROF_FWD_BWD):
    """-> (shortest, longest) launch in milliseconds of one kind since the last reset (reni_profile_minmax; call before the reset)."""
    lo, hi = ctypes.c_double(0.0), ctypes.c_double(0.0)
    _lib.check(_lib.load().reni_profile_minmax(int(kind), ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def profile_read(reset: bool = True, kind: int = PROF_FWD_BWD):
    """-> (summed kernel milliseconds, number of launches) of one kind of launch since the last reset
    (kind: the fused forward+loss+backward pass, RENITestLoss's statistics pass, plain inference, or all)."""
    tot = ctypes.c_double(0.0)
    n = ctypes.c_int64(0)
    _lib.check(_lib.load().reni_profile_read_kind(int(kind), ctypes.byref(tot), ctypes.byref(n), 1 if reset else 0))
    return tot.value, n.value


def _shade_call(forward: bool, normals, positions, camera_center, light_dirs, src, shininess, kd, ks):
    """reni_envmap_shade / reni_envmap_shade_backward (include/reni_hip.h).  normals, positions [NP,3];
    light_dirs [J,3] (shared grid) or [B,J,3]; src = light colours [B,J,3] (forward) or d colours [B,NP,3] (backward)."""
    _require_cuda(normals, positions, light_dirs, src)
    lib = _lib.load()
    f32 = torch.float32
    normals = normals.to(f32).contiguous(); positions = positions.to(f32).contiguous()
    light_dirs = light_dirs.to(f32).contiguous(); src = src.to(f32).contiguous()
    NP = normals.shape[0]
    if normals.shape != (NP, 3) or positions.shape != (NP, 3):
        raise ValueError("normals and positions must be [NP, 3]")
    B = src.shape[0]
    if light_dirs.dim() == 2:
        J, stride = light_dirs.shape[0], 0
    else:
        if light_dirs.shape[0] != B:
            raise ValueError("light_dirs batch size must match the colours'")
        J, stride = light_dirs.shape[1], light_dirs.shape[1] * 3
    want = (B, J, 3) if forward else (B, NP, 3)
    if tuple(src.shape) != want:
        raise ValueError(f"expected a {want} tensor, got {tuple(src.shape)}")
    cam = [float(x) for x in torch.as_tensor(camera_center).reshape(-1)[:3].tolist()]
    out = torch.empty((B, NP, 3) if forward else (B, J, 3), dtype=f32, device=src.device)
    nbytes = int(lib.reni_envmap_shade_workspace_bytes(B, NP, J))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=src.device)
    stream = torch.cuda.current_stream(src.device).cuda_stream
    fn = lib.reni_envmap_shade if forward else lib.reni_envmap_shade_backward
    with torch.cuda.device(src.device):
        _lib.check(fn(B, NP, J, normals.data_ptr(), positions.data_ptr(), cam[0], cam[1], cam[2], light_dirs.data_ptr(),
                      stride, src.data_ptr(), float(shininess), float(kd), float(ks), out.data_ptr(), ws.data_ptr(), nbytes,
                      stream))
    return out


def envmap_shade(normals, positions, camera_center, light_dirs, light_colors, shininess, kd, ks):
    """colors [B,NP,3] of the Blinn-Phong environment-map shader (pytorch3d_envmap_shader.py:75-115)."""
    return _shade_call(True, normals, positions, camera_center, light_dirs, light_colors, shininess, kd, ks)


def envmap_shade_backward(normals, positions, camera_center, light_dirs, dcolors, shininess, kd, ks):
    """d loss / d light_colors [B,J,3] for an upstream d loss / d colors [B,NP,3]."""
    return _shade_call(False, normals, positions, camera_center, light_dirs, dcolors, shininess, kd, ks)


def _ws256(nbytes, device):
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=device)
    p = ws.data_ptr()
    ap = (p + 255) & ~255
    return ws, ap, ws.numel() - (ap - p)


def unnormalise_srgb(img: torch.Tensor, minmax=None, srgb: bool = True, want_linear: bool = False):
    """reni_unnormalise_srgb: UnMinMaxNormlise(minmax) (skipped when minmax is None) -> sRGB view, on the device.
    img: [B,3,H,W] or [3,H,W], ANY strides (e.g. a model output viewed as ``out.view(B,H,W,3).permute(0,3,1,2)`` is read
    in place).  Returns the sRGB batch [B,3,H,W] (srgb=True), the linear HDR batch (srgb=False), or both (want_linear)."""
    _require_cuda(img)
    lib = _lib.load()
    if img.dim() == 3:
        img = img.unsqueeze(0)
    if img.dim() != 4 or img.shape[1] != 3:
        raise ValueError(f"expected [B,3,H,W] or [3,H,W], got {tuple(img.shape)}")
    if img.dtype != torch.float32:
        img = img.float()
    B, _, H, W = img.shape
    st = (ctypes.c_int64 * 4)(*img.stride())
    dev = img.device
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev) if srgb else None
    lin = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev) if (want_linear or not srgb) else None
    ws, wp, wn = _ws256(int(lib.reni_image_workspace_bytes(B, H, W)), dev) if srgb else (None, None, 0)
    m0, m1 = (float(minmax[0]), float(minmax[1])) if minmax is not None else (0.0, 1.0)
    with torch.cuda.device(dev):
        _lib.check(lib.reni_unnormalise_srgb(B, H, W, img.data_ptr(), st, 0 if minmax is None else 1, m0, m1, 1 if srgb else 0,
                                             out.data_ptr() if out is not None else None,
                                             lin.data_ptr() if lin is not None else None, wp, wn,
                                             torch.cuda.current_stream(dev).cuda_stream))
    if srgb and want_linear:
        return out, lin
    return out if srgb else lin


def minmax_normalise(img: torch.Tensor, minmax):
    """reni_minmax_normalise: MinMaxNormalise(minmax) of ONE image (any shape; the clip bounds are the image's own)."""
    _require_cuda(img)
    lib = _lib.load()
    x = _f32c(img)
    out = torch.empty_like(x)
    ws, wp, wn = _ws256(256, x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.reni_minmax_normalise(x.numel(), x.data_ptr(), float(minmax[0]), float(minmax[1]), out.data_ptr(), wp, wn,
                                             torch.cuda.current_stream(x.device).cuda_stream))
    return out


def launch_count(reset: bool = False) -> int:
    """Kernel launches the library has issued so far in this process (reni_launch_count)."""
    return int(_lib.load().reni_launch_count(1 if reset else 0))
