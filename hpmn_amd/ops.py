"""Host-side operators over the C ABI of libhpmn_hip.so.

Everything numerical on the hot path runs in the HIP library; PyTorch supplies device memory, streams
and events.  There is no autograd here: a training step is a fixed sequence of library calls
(``scan_forward_train`` -> ``read_fwd_bwd`` -> ``scan_backward``), each wrapper below is one entry
point of include/hpmn_hip.h.

Reference being replaced: ``Hpmn.embedding`` + ``Hpmn_Basic.build_memory`` (code/hpmn.py:414-430,
:113-129), the read path ``get_covreg`` / ``query_memory`` / ``attention`` / ``build_fc_net``
(:161-207) and TF's autodiff of all of it.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import HpmnGruBwd, HpmnGruFwd, HpmnGruWgrad, HpmnInputProj, HpmnScanDesc


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("expected contiguous float32 CUDA tensor, got %s %s contiguous=%s"
                             % (t.device, t.dtype, t.is_contiguous()))


def _chk_ids(ids):
    if not (ids.is_cuda and ids.dtype in (torch.int32, torch.int64) and ids.is_contiguous()):
        raise ValueError("ids must be a contiguous int32 or int64 CUDA tensor")


ID_MASK0, ID_I64 = 1, 2          # include/hpmn_hip.h: HPMN_ID_MASK0, HPMN_ID_I64


def _idf(ids, mask_id0) -> int:
    """The id-flags word of the C ABI (the slot named mask_id0): bit 0 = id-0 mask, bit 1 = the ids tensor is int64 (tables
    beyond 2^31 - 1 rows; the reference's placeholders are int32, code/hpmn.py:248-251)."""
    return (ID_MASK0 if mask_id0 else 0) | (ID_I64 if ids is not None and ids.dtype == torch.int64 else 0)


@dataclass(frozen=True)
class ScanSpec:
    """Static description of one branch's build_memory (code/hpmn.py:113-131, 284-296, 436-442)."""
    F: int
    E: int
    H: int
    K: int
    T: int                       # user_maxlen as fed by the loader
    periods: Tuple[int, ...]     # li_layer
    front_zero: int = 0          # 23 for Hpmn_Industry (code/hpmn.py:288-290)
    mask_id0: bool = True        # Hpmn (code/hpmn.py:417-422); False for Hpmn_Industry
    last_index: int = -1         # code/hpmn.py:439 (-1) / :292 (-2)

    @property
    def D0(self) -> int:
        return self.F * self.E

    def layer_lengths(self) -> List[int]:
        out, t = [], self.T + self.front_zero
        for i in range(self.K):
            out.append(t)
            if t % self.periods[i] != 0:
                raise ValueError("layer %d: length %d not divisible by period %d (the tf.reshape at "
                                 "code/hpmn.py:124 would fail too)" % (i, t, self.periods[i]))
            t //= self.periods[i]
        return out

    # (r6) HPMN_ID_HOT: the gradient scatter pre-reduces a wave's equal ids in an LDS table before its atomic row adds: the
    # difference between 3.01 and 2.40 ms per C3 step on Zipf(1.1) item ids (12 % of a batch's lookups share ONE row); Taobao's
    # 4-valued btag column: C2 0.93 -> 0.91.  On uniform ids the table only costs (the scatter launch 105 -> 133 us in C3's tail).
    # HPMN_SCATTER_HOT_HINT = auto (default: the model looks at its first training batch, hpmn.py _probe_id_law), 0, 1.
    hot_ids = os.environ.get("HPMN_SCATTER_HOT_HINT", "auto") == "1"

    def id_flags(self, ids) -> int:
        return _idf(ids, self.mask_id0) | (4 if self.hot_ids else 0)

    def desc(self, B: int, V: int, ids=None) -> HpmnScanDesc:
        d = HpmnScanDesc()
        d.B, d.T, d.F, d.E, d.H, d.K = B, self.T, self.F, self.E, self.H, self.K
        d.front_zero, d.mask_id0, d.last_index, d.V = self.front_zero, self.id_flags(ids), self.last_index, V
        for i in range(self.K):
            d.periods[i] = self.periods[i]
        return d


# ---------------------------------------------------------------------------------------
# thin wrappers, one per C entry point
# ---------------------------------------------------------------------------------------
def embed_gather(ids: torch.Tensor, emb: torch.Tensor, mask_id0: bool) -> torch.Tensor:
    """hpmn_embed_gather: ids [..., F] int32 -> [..., F*E]."""
    _chk_ids(ids)
    _chk_f32(emb)
    F = ids.shape[-1]
    V, E = emb.shape
    N = ids.numel() // F
    out = torch.empty(*ids.shape[:-1], F * E, device=emb.device, dtype=torch.float32)
    rc = _lib.load().hpmn_embed_gather(ids.data_ptr(), emb.data_ptr(), out.data_ptr(), N, F, E, V,
                                        _idf(ids, mask_id0), _stream())
    _lib.check(rc, "hpmn_embed_gather")
    return out


def gru_input_proj(spec_or_none, *, x=None, ids=None, emb=None, wg, bg, wc, bc, H, T, front_zero=0,
                   mask_id0=False, want_x_out=False, out=None, t_range=None):
    """hpmn_gru_input_proj.  Returns (xp [B,T,3H], x_out or None).  ``out`` = (xp, x_out) preallocated
    buffers and ``t_range`` = (t_begin, t_len) restrict the launch to a time chunk of every sequence."""
    a = HpmnInputProj()
    _chk_f32(wg, bg, wc, bc)
    if x is not None:
        _chk_f32(x)
        B, Tx, D = x.shape
        assert Tx == T
        a.x = x.data_ptr()
        dev = x.device
    else:
        _chk_ids(ids)
        _chk_f32(emb)
        B, Tids, F = ids.shape
        V, E = emb.shape
        D = F * E
        a.ids, a.emb = ids.data_ptr(), emb.data_ptr()
        a.Tids, a.F, a.E, a.front_zero, a.mask_id0, a.V = Tids, F, E, front_zero, _idf(ids, mask_id0), V
        dev = emb.device
    a.B, a.T, a.D, a.H = B, T, D, H
    a.wg, a.bg, a.wc, a.bc = wg.data_ptr(), bg.data_ptr(), wc.data_ptr(), bc.data_ptr()
    if out is not None:
        xp, x_out = out
    else:
        xp = torch.empty(B, T, 3 * H, device=dev, dtype=torch.float32)
        x_out = torch.empty(B, T, D, device=dev, dtype=torch.float32) if (want_x_out and x is None) else None
    if x_out is not None and x is None:
        a.x_out = x_out.data_ptr()
    if t_range is not None:
        a.t_begin, a.t_len = t_range
    a.xp = xp.data_ptr()
    rc = _lib.load().hpmn_gru_input_proj(C.byref(a), _stream())
    _lib.check(rc, "hpmn_gru_input_proj")
    return xp, x_out


FUSED_FWD = int(os.environ.get("HPMN_FUSED_FWD", "1")) != 0


def fused_fwd_supported(H: int, D: int, gather: bool) -> bool:
    return FUSED_FWD and bool(_lib.load().hpmn_gru_fused_fwd_supported(H, D, int(gather)))


def gru_fused_fwd(*, x=None, ids=None, emb=None, wg, bg, wc, bc, H, T, front_zero=0, mask_id0=False, h_last, period,
                  out, no_candidate=False):
    """hpmn_gru_fused_fwd: input projection + recurrence of one layer in one launch (two specialised waves per
    sequence).  ``out`` = (y, hs, gates, x_out), any of them None.  ``no_candidate``: HPMN_FWD_NO_CANDIDATE (the candidate
    third of ``gates`` stays unwritten; the reverse scan then needs ``candidate_from_hs``)."""
    a = _lib.HpmnGruFusedFwd()
    _chk_f32(wg, bg, wc, bc)
    if x is not None:
        _chk_f32(x)
        B, Tx, D = x.shape
        assert Tx == T
        a.x = x.data_ptr()
    else:
        _chk_ids(ids)
        _chk_f32(emb)
        B, Tids, F = ids.shape
        V, E = emb.shape
        D = F * E
        a.ids, a.emb = ids.data_ptr(), emb.data_ptr()
        a.Tids, a.F, a.E, a.front_zero, a.mask_id0, a.V = Tids, F, E, front_zero, _idf(ids, mask_id0), V
    a.B, a.T, a.D, a.H = B, T, D, H
    a.wg, a.bg, a.wc, a.bc = wg.data_ptr(), bg.data_ptr(), wc.data_ptr(), bc.data_ptr()
    assert h_last.stride(1) == 1 and h_last.shape == (B, H)
    a.h_last, a.h_last_stride = h_last.data_ptr(), h_last.stride(0)
    a.period = period
    y, hs, gates, x_out = out
    a.y, a.hs, a.gates, a.x_out = _ptr(y), _ptr(hs), _ptr(gates), _ptr(x_out)
    a.flags = _lib.HPMN_FWD_NO_CANDIDATE if no_candidate else 0
    rc = _lib.load().hpmn_gru_fused_fwd(C.byref(a), _stream())
    _lib.check(rc, "hpmn_gru_fused_fwd")


def _fill_fused(a, *, x, ids, emb, wg, bg, wc, bc, H, T, front_zero, mask_id0, h_last, period, out, no_candidate=False):
    _chk_f32(wg, bg, wc, bc)
    a.flags = _lib.HPMN_FWD_NO_CANDIDATE if no_candidate else 0
    if x is not None:
        _chk_f32(x)
        B, Tx, D = x.shape
        assert Tx == T
        a.x = x.data_ptr()
    elif ids is not None:
        _chk_ids(ids)
        _chk_f32(emb)
        B, Tids, F = ids.shape
        V, E = emb.shape
        D = F * E
        a.ids, a.emb = ids.data_ptr(), emb.data_ptr()
        a.Tids, a.F, a.E, a.front_zero, a.mask_id0, a.V = Tids, F, E, front_zero, _idf(ids, mask_id0), V
    else:                                   # the upper layer of a pair: its rows never leave the CU
        B, D = h_last.shape[0], H
    a.B, a.T, a.D, a.H = B, T, D, H
    a.wg, a.bg, a.wc, a.bc = wg.data_ptr(), bg.data_ptr(), wc.data_ptr(), bc.data_ptr()
    assert h_last.stride(1) == 1 and h_last.shape == (B, H)
    a.h_last, a.h_last_stride = h_last.data_ptr(), h_last.stride(0)
    a.period = period
    y, hs, gates, x_out = out
    a.y, a.hs, a.gates, a.x_out = _ptr(y), _ptr(hs), _ptr(gates), _ptr(x_out)


_pair_scratch = {}


def gru_pair_fwd(lo: dict, up: dict, flags: int = 0):
    """hpmn_gru_pair_fwd: two consecutive layers in one launch.  ``lo`` / ``up``: keyword arguments of gru_fused_fwd
    (``up`` without x / ids: its input rows are handed over inside the launch)."""
    lib = _lib.load()
    p = _lib.HpmnGruPairFwd()
    _fill_fused(p.lo, x=lo.get("x"), ids=lo.get("ids"), emb=lo.get("emb"), wg=lo["wg"], bg=lo["bg"], wc=lo["wc"],
                bc=lo["bc"], H=lo["H"], T=lo["T"], front_zero=lo.get("front_zero", 0), mask_id0=lo.get("mask_id0", False),
                h_last=lo["h_last"], period=lo["period"], out=lo["out"], no_candidate=lo.get("no_candidate", False))
    _fill_fused(p.up, x=None, ids=None, emb=None, wg=up["wg"], bg=up["bg"], wc=up["wc"], bc=up["bc"], H=up["H"],
                T=up["T"], front_zero=0, mask_id0=False, h_last=up["h_last"], period=up["period"], out=up["out"],
                no_candidate=up.get("no_candidate", False))
    dev = lo["wg"].device
    sc = _pair_scratch.get(str(dev))
    if sc is None:
        sc = _pair_scratch[str(dev)] = torch.empty(int(lib.hpmn_gru_pair_fwd_scratch_bytes()), device=dev, dtype=torch.uint8)
    p.scratch, p.flags = sc.data_ptr(), flags
    _lib.check(lib.hpmn_gru_pair_fwd(C.byref(p), _stream()), "hpmn_gru_pair_fwd")


def gru_scan_fwd(xp, wg, wc, D, h_last, period, want_y, train, out=None, t_range=None, h_init=None):
    """hpmn_gru_scan_fwd.  h_last is a [B,H] strided view (e.g. memory[:, i, :]).  ``out`` = (y, hs, gates)
    preallocated; ``t_range`` = (t_begin, t_end) runs one time chunk from ``h_init`` ([B,H] strided view)."""
    B, T, H3 = xp.shape
    H = H3 // 3
    _chk_f32(xp, wg, wc)
    a = HpmnGruFwd()
    a.B, a.T, a.D, a.H = B, T, D, H
    a.xp, a.wg, a.wc = xp.data_ptr(), wg.data_ptr(), wc.data_ptr()
    assert h_last.stride(1) == 1 and h_last.shape == (B, H)
    a.h_last, a.h_last_stride = h_last.data_ptr(), h_last.stride(0)
    a.period = period
    y = hs = gates = None
    if out is not None:
        y, hs, gates = out
    else:
        if want_y:
            y = torch.empty(B, T // period, H, device=xp.device, dtype=torch.float32)
        if train:
            hs = torch.empty(B, T + 1, H, device=xp.device, dtype=torch.float32)
            gates = torch.empty(B, T, 3 * H, device=xp.device, dtype=torch.float32)
    if y is not None:
        a.y = y.data_ptr()
    if hs is not None:
        a.hs, a.gates = hs.data_ptr(), gates.data_ptr()
    if t_range is not None:
        a.t_begin, a.t_end = t_range
    if h_init is not None:
        assert h_init.stride(1) == 1
        a.h_init, a.h_init_stride = h_init.data_ptr(), h_init.stride(0)
    rc = _lib.load().hpmn_gru_scan_fwd(C.byref(a), _stream())
    _lib.check(rc, "hpmn_gru_scan_fwd")
    return y, hs, gates


def scan_bwd_fuses_dx(H: int, B: int) -> bool:
    return bool(_lib.load().hpmn_gru_scan_bwd_fuses_dx(H, B))


def candidate_elision(H: int, B: int) -> bool:
    """hpmn_gru_candidate_elision: saved gates without the candidate are supported by both directions (ABI v11)."""
    return bool(_lib.load().hpmn_gru_candidate_elision(H, B))


def gru_scan_bwd(wg, wc, D, hs, gates, d_h_last, d_y, period, out=None, t_range=None, dh_carry=None, d_x=None,
                 candidate_from_hs=False):
    """hpmn_gru_scan_bwd -> d_act [B,T,3H].  ``out`` = preallocated d_act; ``t_range`` = (t_begin, t_end)
    runs one time chunk (reverse) with the boundary gradient handed over through ``dh_carry`` [B,H]."""
    B, T1, H = hs.shape
    T = T1 - 1
    _chk_f32(wg, wc, hs, gates, d_y)
    a = HpmnGruBwd()
    a.B, a.T, a.D, a.H = B, T, D, H
    a.wg, a.wc, a.hs, a.gates = wg.data_ptr(), wc.data_ptr(), hs.data_ptr(), gates.data_ptr()
    assert d_h_last.stride(1) == 1 and d_h_last.shape == (B, H) and d_h_last.dtype == torch.float32
    a.d_h_last, a.d_h_last_stride = d_h_last.data_ptr(), d_h_last.stride(0)
    a.d_y = _ptr(d_y)
    a.period = period
    d_act = out if out is not None else torch.empty(B, T, 3 * H, device=hs.device, dtype=torch.float32)
    a.d_act = d_act.data_ptr()
    if t_range is not None:
        a.t_begin, a.t_end = t_range
    if dh_carry is not None:
        a.dh_carry = dh_carry.data_ptr()
    if d_x is not None:             # the input gradient from the scan launch itself (scan_bwd_fuses_dx)
        a.d_x = d_x.data_ptr()
    a.flags = _lib.HPMN_BWD_CANDIDATE_FROM_HS if candidate_from_hs else 0
    rc = _lib.load().hpmn_gru_scan_bwd(C.byref(a), _stream())
    _lib.check(rc, "hpmn_gru_scan_bwd")
    return d_act


def _fill_bwd(a, *, wg, wc, D, hs, gates, d_h_last, d_y, period, d_act, d_x=None, candidate_from_hs=False):
    B, T1, H = hs.shape
    a.flags = _lib.HPMN_BWD_CANDIDATE_FROM_HS if candidate_from_hs else 0
    _chk_f32(wg, wc, hs, gates, d_y, d_act)
    a.B, a.T, a.D, a.H = B, T1 - 1, D, H
    a.wg, a.wc, a.hs, a.gates = wg.data_ptr(), wc.data_ptr(), hs.data_ptr(), gates.data_ptr()
    assert d_h_last.stride(1) == 1 and d_h_last.shape == (B, H) and d_h_last.dtype == torch.float32
    a.d_h_last, a.d_h_last_stride = d_h_last.data_ptr(), d_h_last.stride(0)
    a.d_y, a.period, a.d_act, a.d_x = _ptr(d_y), period, d_act.data_ptr(), _ptr(d_x)


def gru_pair_bwd(lo: dict, up: dict, flags: int = 0):
    """hpmn_gru_pair_bwd: the reverse scans of two consecutive layers in one launch.  ``lo`` / ``up``: wg, wc, D, hs, gates,
    d_h_last, period, d_act (+ ``up['d_y']`` from memory or None, ``lo['d_x']`` optional)."""
    p = _lib.HpmnGruPairBwd()
    _fill_bwd(p.lo, d_y=None, **lo)
    _fill_bwd(p.up, **up)
    p.flags = flags
    _lib.check(_lib.load().hpmn_gru_pair_bwd(C.byref(p), _stream()), "hpmn_gru_pair_bwd")


def gru_param_grads(x, hs, gates, d_act, wg, wc, d_wg, d_bg, d_wc, d_bc, want_dx=True, keep=None, t_range=None,
                    whole_cu=False):
    """hpmn_gru_param_grads: accumulates into d_wg/d_bg/d_wc/d_bc (caller-zeroed), returns dx or None.
    ``keep``: list that receives the temporaries (workspace) when the call is issued on a side stream;
    ``t_range`` = (t_begin, t_len) restricts the reduction to those steps of every sequence."""
    B, T, D = x.shape
    H = hs.shape[2]
    _chk_f32(x, hs, gates, d_act, wg, wc, d_wg, d_bg, d_wc, d_bc)
    a = HpmnGruWgrad()
    a.B, a.T, a.D, a.H = B, T, D, H
    a.x, a.hs, a.gates, a.d_act = x.data_ptr(), hs.data_ptr(), gates.data_ptr(), d_act.data_ptr()
    a.wg, a.wc = wg.data_ptr(), wc.data_ptr()
    a.d_wg, a.d_bg, a.d_wc, a.d_bc = d_wg.data_ptr(), d_bg.data_ptr(), d_wc.data_ptr(), d_bc.data_ptr()
    if t_range is not None:
        a.t_begin, a.t_len = t_range
    a.whole_cu = 1 if whole_cu else 0          # (layer 0 at the end of BPTT: nothing latency-critical beside it)
    d_x = None
    if want_dx:
        d_x = torch.empty(B, T, D, device=x.device, dtype=torch.float32)
        a.d_x = d_x.data_ptr()
    lib = _lib.load()
    ws = torch.empty(lib.hpmn_gru_param_grads_workspace_bytes(B, T, D, H) // 4, device=x.device,
                     dtype=torch.float32)
    if keep is not None:
        keep.append(ws)
    a.workspace = ws.data_ptr()
    rc = lib.hpmn_gru_param_grads(C.byref(a), _stream())
    _lib.check(rc, "hpmn_gru_param_grads")
    return d_x


def gru_input_grad(d_act, wg, wc, D, out=None, t_range=None):
    """hpmn_gru_input_grad: dx [B,T,D] = d_act [Wg[:D] | Wc[:D]]^T (optionally one time chunk)."""
    B, T, H3 = d_act.shape
    _chk_f32(d_act, wg, wc)
    d_x = out if out is not None else torch.empty(B, T, D, device=d_act.device, dtype=torch.float32)
    tb, tl = t_range if t_range is not None else (0, 0)
    rc = _lib.load().hpmn_gru_input_grad(d_act.data_ptr(), wg.data_ptr(), wc.data_ptr(), d_x.data_ptr(),
                                          B, T, D, H3 // 3, tb, tl, _stream())
    _lib.check(rc, "hpmn_gru_input_grad")
    return d_x


def embed_grad_scatter(ids, d_x, d_emb, front_zero, mask_id0):
    """hpmn_embed_grad_scatter: d_emb[ids] += d_x rows (atomic, run-length pre-reduced)."""
    _chk_ids(ids)
    _chk_f32(d_x, d_emb)
    B, T, F = ids.shape
    V, E = d_emb.shape
    assert d_x.shape == (B, front_zero + T, F * E)
    rc = _lib.load().hpmn_embed_grad_scatter(ids.data_ptr(), d_x.data_ptr(), d_emb.data_ptr(), B, T, F, E,
                                              front_zero, V, _idf(ids, mask_id0), _stream())
    _lib.check(rc, "hpmn_embed_grad_scatter")


def adam_step(param, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0, grad_scale=1.0, clear_grad=False):
    """hpmn_adam_step over flat fp32 buffers (code/hpmn.py:209-214).  ``clear_grad``: hpmn_adam_step_clear -- the gradient is
    consumed (left all-zero for the next step)."""
    _chk_f32(param, grad, m, v)
    n = param.numel()
    assert grad.numel() == n and m.numel() == n and v.numel() == n
    fn = _lib.load().hpmn_adam_step_clear if clear_grad else _lib.load().hpmn_adam_step
    rc = fn(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), n, lr_t, beta1, beta2, eps, clip, grad_scale,
            _stream())
    _lib.check(rc, "hpmn_adam_step_clear" if clear_grad else "hpmn_adam_step")


def adam_step_rows(param, grad_rows, m, v, row_ids, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0, grad_scale=1.0):
    """hpmn_adam_step_rows: the update of ``adam_step`` on the table rows ``row_ids`` [U] int64 only, gradients
    from the compact ``grad_rows`` [U, E] (lazy / sparse Adam -- a labelled deviation, see include/hpmn_hip.h)."""
    _chk_f32(param, grad_rows, m, v)
    assert row_ids.dtype == torch.int64 and row_ids.is_cuda and row_ids.is_contiguous()
    U, E = grad_rows.shape
    assert row_ids.numel() == U and param.shape[1] == E
    rc = _lib.load().hpmn_adam_step_rows(param.data_ptr(), grad_rows.data_ptr(), m.data_ptr(), v.data_ptr(),
                                          row_ids.data_ptr(), U, E, lr_t, beta1, beta2, eps, clip, grad_scale, _stream())
    _lib.check(rc, "hpmn_adam_step_rows")


def table_mark_rows(ids: torch.Tensor, flags: torch.Tensor):
    """hpmn_table_mark_rows: flags[id] = 1 for every id of the batch (flags: uint8 [V], all zero before)."""
    _chk_ids(ids)
    assert flags.dtype == torch.uint8 and flags.is_cuda and flags.is_contiguous()
    rc = _lib.load().hpmn_table_mark_rows(ids.data_ptr(), ids.numel(), flags.data_ptr(), flags.numel(), _idf(ids, False),
                                           _stream())
    _lib.check(rc, "hpmn_table_mark_rows")


def adam_step_table(param, grad, m, v, flags, pass_: int, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0,
                    grad_scale=1.0):
    """hpmn_adam_step_table: the dense table update of ``adam_step`` in two passes (0: rows no id of the batch points
    at, gradient taken as zero; 1: the marked rows, whose gradient rows and flags it clears).  [V, E] views."""
    _chk_f32(param, m, v) if grad is None else _chk_f32(param, grad, m, v)
    V, E = param.shape
    assert (grad is None and pass_ == 0 or grad.shape == param.shape) and flags.numel() == V and flags.dtype == torch.uint8
    rc = _lib.load().hpmn_adam_step_table(param.data_ptr(), _ptr(grad), m.data_ptr(), v.data_ptr(), flags.data_ptr(),
                                           V, E, int(pass_), lr_t, beta1, beta2, eps, clip, grad_scale, _stream())
    _lib.check(rc, "hpmn_adam_step_table")


def table_flags(V: int, device) -> torch.Tensor:
    """The byte-per-row flags of the two-pass table Adam: [V] uint8 zeros over an allocation that is a multiple of 4 bytes
    (hpmn_table_mark_ranks ORs rank bits into aligned 32-bit words)."""
    return torch.zeros((V + 3) // 4 * 4, device=device, dtype=torch.uint8)[:V]


class RowBuckets:
    """A bucket index over every rank's ascending row list (hpmn_table_mark_ranks builds it, hpmn_rows_sum_adam searches through
    it): start[r, b] = first entry of list r whose id >> shift is >= b.  ~4 list entries per bucket at the lists' capacity."""

    def __init__(self, world: int, V: int, cap: int, device):
        target = max(1024, int(cap) // 4)
        shift = 0
        while ((V - 1) >> shift) + 1 > target:
            shift += 1
        self.shift, self.nb = shift, ((V - 1) >> shift) + 1
        self.start = torch.empty(world, self.nb + 2, device=device, dtype=torch.int32)


def table_mark_ranks(ids_all: torch.Tensor, counts: Optional[torch.Tensor], flags: torch.Tensor, cap: Optional[int] = None,
                     counts_stride: int = 1, buckets: Optional[RowBuckets] = None):
    """hpmn_table_mark_ranks: flags[row] |= 1 << r for the valid entries of ids_all[r, :] (``counts``: int32 device tensor,
    rank r's list length at counts[r * counts_stride]; None: the first ``cap`` entries, ids outside [0, V) ignored).
    ``buckets``: the same pass fills the bucket index."""
    _chk_ids(ids_all)
    assert ids_all.dim() == 2 and flags.dtype == torch.uint8 and flags.is_cuda and flags.is_contiguous()
    world, stride = ids_all.shape
    if counts is not None:
        assert counts.dtype == torch.int32 and counts.is_cuda and counts.is_contiguous()
    if buckets is not None:
        assert buckets.start.shape[0] == world
    rc = _lib.load().hpmn_table_mark_ranks(ids_all.data_ptr(), stride, world, _ptr(counts), counts_stride,
                                            stride if cap is None else cap, flags.data_ptr(), flags.numel(),
                                            _idf(ids_all, False), _ptr(buckets.start if buckets is not None else None),
                                            buckets.start.shape[1] if buckets is not None else 0,
                                            buckets.shift if buckets is not None else 0, _stream())
    _lib.check(rc, "hpmn_table_mark_ranks")


def rows_sum_adam(param, m, v, flags, ids_all, rows_all, lr_t, *, counts=None, counts_stride=1, lens=None, first=None,
                  n=None, beta1=0.9, beta2=0.999, eps=1e-8, clip=1.0, grad_scale=1.0, buckets: Optional[RowBuckets] = None):
    """hpmn_rows_sum_adam: the update of the TOUCHED table rows from compact gradient rows, all ranks' lists in one launch.
    ``ids_all`` [world, ids_stride] (every rank's ascending distinct rows), ``rows_all`` [world, rows_stride, E] (their
    gradient rows; rows_all[r][i] belongs to list entry first[r] + i), list lengths from the device tensor ``counts`` (int32,
    rank r at counts[r * counts_stride]) or the host list ``lens``; the call consumes the window [first[r], first[r] + n[r])
    of every list (default: everything up to ids_stride).  param / m / v: [V, E] views; flags: the rank-bit bytes."""
    _chk_f32(param, m, v, rows_all)
    _chk_ids(ids_all)
    assert ids_all.dim() == 2 and rows_all.dim() == 3 and rows_all.shape[0] == ids_all.shape[0]
    world, ids_stride = ids_all.shape
    V, E = param.shape
    assert rows_all.shape[2] == E and flags.numel() == V and flags.dtype == torch.uint8
    a = _lib.HpmnRowsAdam()
    a.world, a.E, a.id_flags, a.counts_stride = world, E, _idf(ids_all, False), counts_stride
    a.ids, a.ids_stride = ids_all.data_ptr(), ids_stride
    if counts is not None:
        assert counts.dtype == torch.int32 and counts.is_cuda and counts.is_contiguous()
        a.counts = counts.data_ptr()
    for r in range(world):
        a.len[r] = int(lens[r]) if lens is not None else ids_stride
        a.first[r] = int(first[r]) if first is not None else 0
        a.n[r] = int(n[r]) if n is not None else min(ids_stride, rows_all.shape[1]) - a.first[r]
    a.rows, a.rows_stride = rows_all.data_ptr(), rows_all.shape[1]
    a.flags, a.param, a.m, a.v, a.V = flags.data_ptr(), param.data_ptr(), m.data_ptr(), v.data_ptr(), V
    a.lr_t, a.beta1, a.beta2, a.eps, a.clip, a.grad_scale = lr_t, beta1, beta2, eps, clip, grad_scale
    if buckets is not None and world > 1:
        a.bucket_start, a.bucket_stride, a.bucket_shift = buckets.start.data_ptr(), buckets.start.shape[1], buckets.shift
    _lib.check(_lib.load().hpmn_rows_sum_adam(C.byref(a), _stream()), "hpmn_rows_sum_adam")


def scan_forward_inference(spec: ScanSpec, ids, emb, weights: Sequence[torch.Tensor], workspace=None):
    """hpmn_scan_fwd: whole build_memory forward (no saved states).  weights = [wg0,bg0,wc0,bc0, wg1,...].
    Returns (memory [B,K,H], last [B,D0])."""
    _chk_ids(ids)
    _chk_f32(emb, *weights)
    B = ids.shape[0]
    V = emb.shape[0]
    lib = _lib.load()
    if B == 0:       # nothing to launch (and a 0-element tensor has a null data pointer)
        return (torch.empty(0, spec.K, spec.H, device=emb.device), torch.empty(0, spec.D0, device=emb.device))
    d = spec.desc(B, V, ids)
    need = lib.hpmn_scan_workspace_bytes(C.byref(d))
    if need == 0:
        raise ValueError("inconsistent scan description (layer lengths must divide by the periods)")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=emb.device, dtype=torch.uint8)
    K = spec.K
    arr = lambda i0: (C.c_void_p * K)(*[weights[4 * i + i0].data_ptr() for i in range(K)])
    memory = torch.empty(B, K, spec.H, device=emb.device, dtype=torch.float32)
    last = torch.empty(B, spec.D0, device=emb.device, dtype=torch.float32)
    rc = lib.hpmn_scan_fwd(C.byref(d), ids.data_ptr(), emb.data_ptr(), arr(0), arr(1), arr(2), arr(3),
                           memory.data_ptr(), last.data_ptr(), workspace.data_ptr(), _stream())
    _lib.check(rc, "hpmn_scan_fwd")
    return memory, last


# ---------------------------------------------------------------------------------------
# build_memory with saved states (training) and its BPTT -- plain kernel sequences, no autograd
# ---------------------------------------------------------------------------------------
import os as _os

# Time chunks for cross-layer pipelining over K HIP streams (1 = off, the default).  Measured on MI355X at
# C3: a cross-stream event dependency costs ~40 us, a chunked scan launch ~15 us of prologue, so 2/4/8
# chunks give 6.05/6.84/8.50 ms per step against 5.45 ms unpipelined -- kept (and tested) as an option.
PIPELINE_CHUNKS = int(_os.environ.get("HPMN_PIPELINE_CHUNKS", "1"))

_layer_streams = {}


def _streams(device, n):
    """n helper streams private to the CURRENT stream (sub-batches run this code concurrently)."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    lst = _layer_streams.setdefault(key, [])
    while len(lst) < n:
        lst.append(torch.cuda.Stream(device=device))
    return lst[:n]


def chunk_plan(spec: ScanSpec, nc: int):
    """Per-layer chunk lengths for cross-layer pipelining, or None if the layer lengths do not split:
    chunk c of layer i = steps [c*L_i, (c+1)*L_i) with L_{i+1} = L_i / period_i; every L_i must be even
    (the kernels stage 2-step chunks) and a multiple of the layer's period."""
    if nc <= 1 or spec.K < 2:
        return None
    lens = spec.layer_lengths()
    if lens[0] % nc:
        return None
    L, out = lens[0] // nc, []
    for i in range(spec.K):
        if L < 2 or L % 2 or lens[i] != L * nc:
            return None
        out.append(L)
        if i + 1 < spec.K:
            if L % spec.periods[i]:
                return None
            L //= spec.periods[i]
    return out


# ---------------------------------------------------------------------------------------
# all layers in one launch (H = 64): hpmn_pipe_fwd / hpmn_pipe_bwd
# ---------------------------------------------------------------------------------------
# HPMN_PIPE = "0" (default): per-layer, per-sequence kernels; "all": every layer in ONE pipelined, batch-tiled MFMA
# launch (hpmn_pipe_fwd / hpmn_pipe_bwd); "upper": layer 0 on the per-sequence kernels, layers 1..K-1 pipelined.
# The pipelined launches are parity-green but measured SLOWER at the reference batch (C3, B=500: forward 1.52 /
# 1.42 / 1.14 ms, backward 2.31 / 2.29 / 2.29 ms for all / upper / 0) -- a tile of 16 sequences concentrates on
# one CU the LDS traffic, transcendental work and stores that one-sequence-per-wave spreads over eight;
# DESIGN_HISTORY.md section 3.7 has the per-step cycle accounting.
PIPE = os.environ.get("HPMN_PIPE", "0")
if PIPE in ("1", "true"):
    PIPE = "all"
_pipe_sync = {}


def pipe_mode(spec: ScanSpec) -> str:
    """"all" / "upper" / "" (per-layer kernels) for this graph."""
    if PIPE in (False, None, "0", "", "off") or spec.H != 64 or spec.E % 4:
        return ""
    lib = _lib.load()
    if not lib.hpmn_has_legacy_kernels():       # (the tile kernel's TRAINING backward is a -DHPMN_LEGACY_KERNELS build)
        return ""
    if PIPE in (True, "all") and lib.hpmn_pipe_supported(spec.H, spec.D0):
        return "all"
    if spec.K >= 3 and lib.hpmn_pipe_supported(spec.H, spec.H):
        return "upper" if PIPE == "upper" else ("upper" if not lib.hpmn_pipe_supported(spec.H, spec.D0) else "all")
    return ""


def pipe_supported(spec: ScanSpec) -> bool:
    return pipe_mode(spec) == "all"


def tile_kernel_supported(H: int, D0: int) -> bool:
    """hpmn_pipe_supported / hpmn_tile_supported: the 16-sequence-tile MFMA scan has instantiations for (H, D0) -- whatever
    the HPMN_PIPE switch says about using it for TRAINING (ops.tiled_forward_inference is the forward-only use)."""
    lib = _lib.load()
    if H == 128:
        return bool(lib.hpmn_tile_supported(H, D0) and TILE128)
    return bool(lib.hpmn_pipe_supported(H, D0) and lib.hpmn_pipe_supported(H, H))


TILE128 = os.environ.get("HPMN_TILE128", "1") != "0"      # evaluation at H = 128 on the tile kernel (0: per-sequence scans)
_cus = {}


def _cu_count(dev) -> int:
    k = str(dev)
    if k not in _cus:
        _cus[k] = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    return _cus[k]


TILE64 = os.environ.get("HPMN_TILE64", "1") != "0"        # evaluation at H = 64 on the four-wave tile kernel (0: gru_pipe_fwd_kernel)
TILE128_XP = os.environ.get("HPMN_TILE128_XP", "0") == "1"   # its upper layers on rows projected by hpmn_gru_input_proj (first version)


def _pipe_sync_buffer(K: int, B: int, device) -> torch.Tensor:
    """Progress words of the in-launch hand-offs, private to the current stream (re-zeroed by every call)."""
    need = _lib.load().hpmn_pipe_sync_bytes(K, B)
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    buf = _pipe_sync.get(key)
    if buf is None or buf.numel() < need:
        buf = _pipe_sync[key] = torch.zeros(need, device=device, dtype=torch.uint8)
    return buf


def embed_gather_sum(ids: torch.Tensor, emb: torch.Tensor, mask_id0: bool, out=None) -> torch.Tensor:
    """hpmn_embed_gather_sum: out[b, f*E:(f+1)*E] = sum_t emb[ids[b,t,f]] -- the gather consumed in place."""
    _chk_ids(ids)
    _chk_f32(emb)
    B, T, F = ids.shape
    V, E = emb.shape
    if out is None:
        out = torch.zeros(B, F * E, device=emb.device, dtype=torch.float32)
    rc = _lib.load().hpmn_embed_gather_sum(ids.data_ptr(), emb.data_ptr(), out.data_ptr(), B, T, F, E, V, _idf(ids, mask_id0),
                                            _stream())
    _lib.check(rc, "hpmn_embed_gather_sum")
    return out


def embed_gather_seq(ids, emb, front_zero: int, mask_id0: bool, out=None):
    """hpmn_embed_gather_seq: ids [B,T,F] -> x0 [B, front_zero+T, F*E] (zero prefix, id-0 mask)."""
    _chk_ids(ids)
    _chk_f32(emb)
    B, T, F = ids.shape
    V, E = emb.shape
    if out is None:
        out = torch.empty(B, front_zero + T, F * E, device=emb.device, dtype=torch.float32)
    rc = _lib.load().hpmn_embed_gather_seq(ids.data_ptr(), emb.data_ptr(), out.data_ptr(), B, T, F, E, front_zero, V,
                                            _idf(ids, mask_id0), _stream())
    _lib.check(rc, "hpmn_embed_gather_seq")
    return out


def _pipe_desc(spec: ScanSpec, B: int, weights, train: bool, first: int = 0):
    """HpmnPipe for layers first..K-1 of the graph (pipe layer j = graph layer first + j)."""
    p = _lib.HpmnPipe()
    lens = spec.layer_lengths()
    p.B, p.K, p.H, p.train = B, spec.K - first, spec.H, int(train)
    p.mem_stride = spec.K * spec.H
    for j, i in enumerate(range(first, spec.K)):
        p.T[j], p.D[j], p.period[j] = lens[i], (spec.D0 if i == 0 else spec.H), spec.periods[i]
        p.wg[j], p.bg[j], p.wc[j], p.bc[j] = (t.data_ptr() for t in weights[4 * i:4 * i + 4])
    return p, lens


def pipe_forward(spec: ScanSpec, ids, emb, weights: Sequence[torch.Tensor], train: bool):
    """build_memory forward with all K layers in ONE launch (hpmn_embed_gather_seq + hpmn_pipe_fwd).
    Returns (memory, last, saved) with saved = [(x_in, hs, gates)] per layer (hs/gates None for inference)."""
    _chk_ids(ids)
    _chk_f32(emb, *weights)
    B = ids.shape[0]
    H, K = spec.H, spec.K
    dev = emb.device
    f32 = dict(device=dev, dtype=torch.float32)
    p, lens = _pipe_desc(spec, B, weights, train)
    x0 = embed_gather_seq(ids, emb, spec.front_zero, spec.mask_id0)
    memory = torch.empty(B, K, H, **f32)
    y = [torch.empty(B, lens[i] // spec.periods[i], H, **f32) if i + 1 < K else None for i in range(K)]
    hs = [torch.empty(B, lens[i] + 1, H, **f32) if train else None for i in range(K)]
    gates = [torch.empty(B, lens[i], 3 * H, **f32) if train else None for i in range(K)]
    p.x0, p.memory = x0.data_ptr(), memory.data_ptr()
    for i in range(K):
        p.y[i], p.hs[i], p.gates[i] = _ptr(y[i]), _ptr(hs[i]), _ptr(gates[i])
    sync = _pipe_sync_buffer(K, B, dev)
    p.sync = sync.data_ptr()
    rc = _lib.load().hpmn_pipe_fwd(C.byref(p), _stream())
    _lib.check(rc, "hpmn_pipe_fwd")
    last = x0[:, spec.last_index, :].contiguous()
    x_in = [x0] + y[:-1]
    return memory, last, [(x_in[i], hs[i], gates[i]) for i in range(K)]


def tiled_forward_inference(spec: ScanSpec, ids, emb, weights: Sequence[torch.Tensor], group: int = 1):
    """build_memory forward for LARGE batches (evaluation): the 16-sequence-tile kernel of hpmn_pipe_fwd -- a step's
    recurrent product as a real [3H x H] x [H x 16] contraction on the matrix cores, split-f16 operands, three products per
    tile, fp32 accumulate (DESIGN_HISTORY.md 3.7) -- run LAYER GROUP BY LAYER GROUP instead of all K layers in one launch.

    Per tile-step the tiled kernel costs ~1830 cycles for 16 sequences where the one-sequence-per-wave kernels cost ~1240
    for (at best) 4-8 per CU, so it wins as soon as the batch fills the chip with tiles; what lost at B = 500 was (a) 32
    tiles on 256 CUs and (b) the in-launch layer pipeline: with all K layers resident, layers 1..K-1 hold a CU each for the
    whole 1024-step duration of layer 0 while doing 1/2, 1/4, ... of its work.  ``group`` layers per launch (1: none of
    that waiting; 2: for batches that only half-fill the chip with tiles) chained through the y rows in memory.
    Returns (memory [B,K,H], last [B,D0]) like scan_forward_inference."""
    _chk_ids(ids)
    _chk_f32(emb, *weights)
    B, H, K = ids.shape[0], spec.H, spec.K
    dev = emb.device
    f32 = dict(device=dev, dtype=torch.float32)
    lens = spec.layer_lengths()
    memory = torch.empty(B, K, H, **f32)
    x = embed_gather_seq(ids, emb, spec.front_zero, spec.mask_id0)
    last = x[:, spec.last_index, :].contiguous()
    lib = _lib.load()
    # (H = 64: the four-wave kernel pays from two tiles per CU -- C3 shape, build_memory: 2.2 M sequences/s at 8 192 rows against
    #  1.6 M on the twelve-wave kernel, but 1.48 M against 1.57 M at 4 096, where every CU holds one tile and nothing interleaves)
    many_tiles = (B + 15) // 16 > _cu_count(dev)
    if H == 128 or (H == 64 and TILE64 and many_tiles and lib.hpmn_tile_supported(64, spec.D0)):
        # r5: hpmn_tile_fwd -- four waves per 16-sequence tile, each with a quarter of the units of ALL THREE gates (r, u and the
        # state stay in its registers), ONE layer per launch, the rows projected in the kernel.  H = 128 (gru_tile128.hip) and,
        # second generation of the H = 64 evaluation kernel, gru_tile64.hip (two or three tiles share a CU)
        for i in range(K):
            wg, bg, wc, bc = weights[4 * i:4 * i + 4]
            a = _lib.HpmnTileFwd()
            a.B, a.T, a.D, a.H, a.period = B, lens[i], (spec.D0 if i == 0 else H), H, spec.periods[i]
            if i == 0 or not TILE128_XP or H == 64:
                # (layers >= 1, r5 second version: the 128-wide rows projected in the kernel too -- hi halves of the input weights
                #  in LDS, lo halves in registers -- instead of 3.5 KB per row-step of projected rows through HBM)
                a.x = x.data_ptr()
                src = x
            else:
                src, _ = gru_input_proj(None, x=x, wg=wg, bg=bg, wc=wc, bc=bc, H=H, T=lens[i])
                a.xp = src.data_ptr()
            a.wg, a.bg, a.wc, a.bc = wg.data_ptr(), bg.data_ptr(), wc.data_ptr(), bc.data_ptr()
            y = torch.empty(B, lens[i] // spec.periods[i], H, **f32) if i + 1 < K else None
            a.y = _ptr(y)
            a.h_last, a.h_last_stride = memory[:, i, :].data_ptr(), K * H
            _lib.check(lib.hpmn_tile_fwd(C.byref(a), _stream()), "hpmn_tile_fwd")
            del src
            x = y
        return memory, last
    first = 0
    while first < K:
        n = min(group, K - first)
        p = _lib.HpmnPipe()
        p.B, p.K, p.H, p.train = B, n, H, 0
        p.mem_stride = K * H
        y = []
        for j in range(n):
            i = first + j
            p.T[j], p.D[j], p.period[j] = lens[i], (spec.D0 if i == 0 else H), spec.periods[i]
            p.wg[j], p.bg[j], p.wc[j], p.bc[j] = (t.data_ptr() for t in weights[4 * i:4 * i + 4])
            y.append(torch.empty(B, lens[i] // spec.periods[i], H, **f32) if i + 1 < K else None)
            p.y[j] = _ptr(y[j])
        p.x0, p.memory = x.data_ptr(), memory[:, first, :].data_ptr()
        p.sync = _pipe_sync_buffer(n, B, dev).data_ptr()
        _lib.check(lib.hpmn_pipe_fwd(C.byref(p), _stream()), "hpmn_pipe_fwd")
        x = y[-1]
        first += n
    return memory, last


def pipe_error_word(K: int, B: int, device) -> int:
    """Hand-off error word of the last forward pipe launch on the current stream (0 = no lost hand-off: a wait
    that exceeds its spin bound gives up, records 1 + layer here and lets the launch finish); synchronises."""
    buf = _pipe_sync_buffer(K, B, device)
    torch.cuda.current_stream().synchronize()
    return int(buf.view(torch.int32)[1])


def pipe_backward(spec: ScanSpec, ids, saved, weights, d_memory, d_last, grad_out, defer_join: bool = False):
    """BPTT of pipe_forward: hpmn_pipe_bwd (every layer's reverse scan + the inter-layer input gradients in one
    launch), then layer 0's input gradient + the embedding scatter on the current stream and the weight
    gradients (MFMA reductions over d_act) on a side stream."""
    K, H = spec.K, spec.H
    B = d_memory.shape[0]
    dev = d_memory.device
    f32 = dict(device=dev, dtype=torch.float32)
    d_emb, gw = grad_out[0], list(grad_out[1:])
    p, lens = _pipe_desc(spec, B, weights, True)
    in_dims = [spec.D0] + [H] * (K - 1)
    d_act = [torch.empty(B, lens[i], 3 * H, **f32) for i in range(K)]
    d_x = [torch.empty(B, lens[i], in_dims[i], **f32) for i in range(K)]
    _chk_f32(d_memory)
    p.x0, p.d_memory = saved[0][0].data_ptr(), d_memory.data_ptr()
    for i in range(K):
        p.hs[i], p.gates[i] = saved[i][1].data_ptr(), saved[i][2].data_ptr()
        p.d_act[i], p.d_x[i] = d_act[i].data_ptr(), d_x[i].data_ptr()
        if i + 1 < K:
            p.y[i] = saved[i + 1][0].data_ptr()
    sync = _pipe_sync_buffer(K, B, dev)
    p.sync = sync.data_ptr()
    rc = _lib.load().hpmn_pipe_bwd(C.byref(p), _stream())
    _lib.check(rc, "hpmn_pipe_bwd")
    main = torch.cuda.current_stream()
    side = _streams(dev, 1)[0]
    keep = []
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for i in range(K):
            x_in, hs, gates = saved[i]
            gru_param_grads(x_in, hs, gates, d_act[i], weights[4 * i], weights[4 * i + 2], gw[4 * i], gw[4 * i + 1],
                            gw[4 * i + 2], gw[4 * i + 3], want_dx=False, keep=keep)
    gru_input_grad(d_act[0], weights[0], weights[2], in_dims[0], out=d_x[0])
    d_x0 = d_x[0]
    d_x0[:, spec.last_index, :] += d_last
    embed_grad_scatter(ids, d_x0, d_emb, spec.front_zero, spec.mask_id0)
    pending = PendingGrads([side], (keep, d_act, d_x, saved))
    if defer_join:
        return pending
    pending.join()
    return None


def upper_forward(spec: ScanSpec, ids, emb, weights: Sequence[torch.Tensor]):
    """Layer 0 on the per-sequence kernels (fused projection + scan), layers 1..K-1 in one pipelined launch."""
    lens = spec.layer_lengths()
    B, H, K = ids.shape[0], spec.H, spec.K
    dev = emb.device
    f32 = dict(device=dev, dtype=torch.float32)
    memory = torch.empty(B, K, H, **f32)
    y = [torch.empty(B, lens[i] // spec.periods[i], H, **f32) if i + 1 < K else None for i in range(K)]
    hs = [torch.empty(B, lens[i] + 1, H, **f32) for i in range(K)]
    gates = [torch.empty(B, lens[i], 3 * H, **f32) for i in range(K)]
    x0 = torch.empty(B, lens[0], spec.D0, **f32)
    wg, bg, wc, bc = weights[0:4]
    if fused_fwd_supported(H, spec.D0, True) and 64 % spec.E == 0:
        gru_fused_fwd(ids=ids, emb=emb, wg=wg, bg=bg, wc=wc, bc=bc, H=H, T=lens[0], front_zero=spec.front_zero,
                      mask_id0=spec.mask_id0, h_last=memory[:, 0, :], period=spec.periods[0],
                      out=(y[0], hs[0], gates[0], x0))
    else:
        xp, _ = gru_input_proj(None, ids=ids, emb=emb, wg=wg, bg=bg, wc=wc, bc=bc, H=H, T=lens[0],
                               front_zero=spec.front_zero, mask_id0=spec.mask_id0,
                               out=(torch.empty(B, lens[0], 3 * H, **f32), x0))
        gru_scan_fwd(xp, wg, wc, spec.D0, memory[:, 0, :], spec.periods[0], True, True, out=(y[0], hs[0], gates[0]))
    p, _ = _pipe_desc(spec, B, weights, True, first=1)
    p.x0, p.memory = y[0].data_ptr(), memory[:, 1, :].data_ptr()
    for j, i in enumerate(range(1, K)):
        p.y[j], p.hs[j], p.gates[j] = _ptr(y[i]), hs[i].data_ptr(), gates[i].data_ptr()
    p.sync = _pipe_sync_buffer(K, B, dev).data_ptr()
    _lib.check(_lib.load().hpmn_pipe_fwd(C.byref(p), _stream()), "hpmn_pipe_fwd")
    last = x0[:, spec.last_index, :].contiguous()
    x_in = [x0] + y[:-1]
    return memory, last, [(x_in[i], hs[i], gates[i]) for i in range(K)]


def upper_backward(spec: ScanSpec, ids, saved, weights, d_memory, d_last, grad_out, defer_join: bool = False):
    """BPTT of upper_forward: layers K-1..1 in one pipelined launch, then layer 0's reverse scan on the
    per-sequence kernel; weight gradients on a side stream as their d_act completes."""
    K, H = spec.K, spec.H
    B = d_memory.shape[0]
    dev = d_memory.device
    f32 = dict(device=dev, dtype=torch.float32)
    d_emb, gw = grad_out[0], list(grad_out[1:])
    lens = spec.layer_lengths()
    in_dims = [spec.D0] + [H] * (K - 1)
    d_act = [torch.empty(B, lens[i], 3 * H, **f32) for i in range(K)]
    d_x = [torch.empty(B, lens[i], in_dims[i], **f32) for i in range(K)]
    _chk_f32(d_memory)
    p, _ = _pipe_desc(spec, B, weights, True, first=1)
    p.x0, p.d_memory = saved[1][0].data_ptr(), d_memory[:, 1, :].data_ptr()
    for j, i in enumerate(range(1, K)):
        p.hs[j], p.gates[j] = saved[i][1].data_ptr(), saved[i][2].data_ptr()
        p.d_act[j], p.d_x[j] = d_act[i].data_ptr(), d_x[i].data_ptr()
        if i + 1 < K:
            p.y[j] = saved[i + 1][0].data_ptr()
    p.sync = _pipe_sync_buffer(K, B, dev).data_ptr()
    _lib.check(_lib.load().hpmn_pipe_bwd(C.byref(p), _stream()), "hpmn_pipe_bwd")
    main = torch.cuda.current_stream()
    side = _streams(dev, 1)[0]
    keep = []

    def wgrad(i):
        x_in, hs, gates = saved[i]
        gru_param_grads(x_in, hs, gates, d_act[i], weights[4 * i], weights[4 * i + 2], gw[4 * i], gw[4 * i + 1],
                        gw[4 * i + 2], gw[4 * i + 3], want_dx=False, keep=keep)

    side.wait_stream(main)
    with torch.cuda.stream(side):
        for i in range(K - 1, 0, -1):
            wgrad(i)
    # the gradient wrt layer 0's subsampled outputs, then its reverse scan (the end of the chain)
    gru_input_grad(d_act[1], weights[4], weights[6], H, out=d_x[1])
    gru_scan_bwd(weights[0], weights[2], in_dims[0], saved[0][1], saved[0][2], d_memory[:, 0, :], d_x[1],
                 spec.periods[0], out=d_act[0])
    side.wait_stream(main)
    with torch.cuda.stream(side):
        wgrad(0)
    gru_input_grad(d_act[0], weights[0], weights[2], in_dims[0], out=d_x[0])
    d_x0 = d_x[0]
    d_x0[:, spec.last_index, :] += d_last
    embed_grad_scatter(ids, d_x0, d_emb, spec.front_zero, spec.mask_id0)
    pending = PendingGrads([side], (keep, d_act, d_x, saved))
    if defer_join:
        return pending
    pending.join()
    return None


# ---------------------------------------------------------------------------------------
# the whole training graph behind two C entry points (hpmn_scan_fwd_train / hpmn_scan_bwd): the default
# ---------------------------------------------------------------------------------------
TRAIN_ABI = int(os.environ.get("HPMN_TRAIN_ABI", "1")) != 0
_train_ctx = {}


def _ctx(device) -> int:
    """One HpmnTrainCtx (helper stream + events) per (device, launch stream)."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    h = _train_ctx.get(key)
    if h is None:
        out = C.c_void_p()
        _lib.check(_lib.load().hpmn_train_ctx_create(C.byref(out)), "hpmn_train_ctx_create")
        h = _train_ctx[key] = out.value
    return h


FAST_PLAN = os.environ.get("HPMN_FAST_PLAN", "1") != "0"      # hpmn_scatter_plan_build instead of torch.sort & co
CHECK_IDS = os.environ.get("HPMN_CHECK_IDS", "0") == "1"      # range-check every plan's ids (blocking; debugging)
_row_bounds = {}


def _row_bounds_tensor(bounds, dev, dtype):
    key = (bounds, str(dev), dtype)
    t = _row_bounds.get(key)
    if t is None:
        if len(_row_bounds) > 64:
            _row_bounds.clear()
        t = _row_bounds[key] = torch.as_tensor(list(bounds), device=dev, dtype=dtype)
    return t


class ScatterPlan:
    """Row order of a batch's lookups (hpmn_scatter_plan): what the deterministic scatter walks.  Built from the ids alone,
    on whatever stream is current (the data-parallel / two-pass step: the auxiliary stream, underneath the forward).
    ``rows[:U]`` = the batch's distinct table rows (ascending), ``out_rows[:U]`` = their gradient rows once the scatter has
    run, ``count`` = U on the device; ``count_host()`` is an event-guarded read of its pinned copy (no device sync)."""

    def __init__(self, ids: torch.Tensor, E: int, want_rows: bool = False, host_count: bool = False, row_bounds=None,
                 rows_capacity: int = 0, out_rows_capacity: int = 0, V: Optional[int] = None):
        _chk_ids(ids)
        dev = ids.device
        flat = ids.reshape(-1)
        n = flat.numel()
        self.n, self.E, self.id_flags = n, E, _idf(ids, False)
        i32 = dict(device=dev, dtype=torch.int32)
        lib = _lib.load()
        # ``rows_capacity`` / ``out_rows_capacity`` (r5): the buffers as the data-parallel exchange sends them -- `rows` is
        # all-gathered whole (the batch geometry's capacity, the same on every rank), slices of `out_rows` go out chunk by
        # chunk, each as long as the LARGEST rank's chunk (what lies behind this rank's own entries is never read)
        nr = max(n, int(rows_capacity))
        self.sorted = self.workspace = self.counts_vec = self.chunk_counts = None
        self.count = torch.zeros(1, **i32)
        nb = len(row_bounds) - 1 if row_bounds is not None else 0
        if FAST_PLAN and n > 0 and nb <= _lib.HPMN_MAX_CHUNKS:
            # r5: one library call -- radix sort of (id, lookup) pairs over the bits the table needs, segment scan, plan
            # kernel, per-chunk counts: ~10 launches where the torch ops below are ~45 (240 us of queue latency at C3)
            vmax = int(V) if V else (2 ** 62 if ids.dtype == torch.int64 else 2 ** 31 - 1)
            if V and CHECK_IDS:
                # the radix sort covers the low key_bits(V) bits only -- an id outside [0, V) is mis-ordered and the
                # plan's `rows` are then neither ascending nor distinct (torch.sort tolerated it).  Staged datasets are
                # range-checked once (_DeviceDataset); a caller's own tensors are checked here under HPMN_CHECK_IDS=1
                # (a device-to-host round trip per plan: a debugging switch)
                lo, hi = int(flat.min()), int(flat.max())
                if lo < 0 or hi >= int(V):
                    raise ValueError("ids outside [0, %d): min %d, max %d" % (int(V), lo, hi))
            need = int(lib.hpmn_scatter_plan_build_workspace_bytes(n, self.id_flags, vmax))
            if need <= 0:
                raise _lib.HpmnLibraryError("hpmn_scatter_plan_build_workspace_bytes refused n=%d V=%d" % (n, vmax))
            self.workspace = torch.empty(need, device=dev, dtype=torch.uint8)
            self.perm, self.seg = torch.empty(n, **i32), torch.empty(n, **i32)
            self.start = torch.empty(n + 1, **i32)
            self.rows = torch.empty(nr, device=dev, dtype=ids.dtype)
            bounds = None
            if row_bounds is not None:
                self.counts_vec = torch.empty(1 + nb, **i32)          # [U, distinct rows per chunk of the table's row range]
                self.chunk_counts = self.counts_vec[1:]
                bounds = (C.c_int64 * (nb + 1))(*[int(x) for x in row_bounds])
            _lib.check(lib.hpmn_scatter_plan_build(flat.data_ptr(), self.id_flags, n, vmax, self.workspace.data_ptr(), need,
                                                   self.perm.data_ptr(), self.seg.data_ptr(), self.start.data_ptr(),
                                                   self.rows.data_ptr(), self.count.data_ptr(), bounds, nb,
                                                   _ptr(self.counts_vec), _stream()), "hpmn_scatter_plan_build")
        else:
            self.sorted, perm = torch.sort(flat, stable=True)
            self.perm = perm.to(torch.int32)
            head = torch.ones(n, **i32)
            if n > 1:
                head[1:] = (self.sorted[1:] != self.sorted[:-1]).to(torch.int32)
            self.seg = torch.cumsum(head, 0, dtype=torch.int32) - 1
            self.start = torch.empty(n + 1, **i32)
            # ``row_bounds`` (ascending table-row boundaries b_0 = 0 < ... < b_C = V): chunk_counts[c] = distinct rows in
            # [b_c, b_c+1) -- the data-parallel exchange sends the rows chunk by chunk (hpmn.py).  The unused tail of `rows` is
            # then filled with the id type's maximum so that a searchsorted over the whole buffer stops at the count.
            self.rows = (torch.full((nr,), torch.iinfo(ids.dtype).max, device=dev, dtype=ids.dtype) if row_bounds is not None
                         else torch.empty(nr, device=dev, dtype=ids.dtype))
            _lib.check(lib.hpmn_scatter_plan(self.sorted.data_ptr(), self.id_flags, n, self.seg.data_ptr(), self.start.data_ptr(),
                                             self.rows.data_ptr(), self.count.data_ptr(), _stream()), "hpmn_scatter_plan")
            if row_bounds is not None:
                # (the boundaries are the same every step: their device tensor is made ONCE.  torch.as_tensor(list, device=...)
                #  here was a blocking pageable host-to-device copy on the auxiliary stream -- which had just been told to wait for
                #  the launch stream, i.e. for the whole previous step: the host could not run ahead of the device any more and
                #  every data-parallel step started ~1.1 ms late, r4: 3.6 -> 2.9 ms per step with one rank on RCCL)
                b = _row_bounds_tensor(tuple(int(x) for x in row_bounds), dev, ids.dtype)
                pos = torch.searchsorted(self.rows, b)                     # first entry >= b_c: [C + 1], pos[C] = U
                self.chunk_counts = (pos[1:] - pos[:-1]).to(torch.int32)
                self.counts_vec = torch.cat([self.count, self.chunk_counts])
        self.partials = torch.empty(max(1, lib.hpmn_embed_grad_segsum_partials_floats(n, E)), device=dev, dtype=torch.float32)
        self.out_rows = (torch.empty(max(n, 1, int(out_rows_capacity)), E, device=dev, dtype=torch.float32)
                         if want_rows else None)
        self._host = self._event = None
        if host_count:
            self._host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            self._host.copy_(self.count, non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()

    def count_host(self) -> int:
        if self._host is None:
            return int(self.count.item())
        self._event.synchronize()
        return int(self._host[0])

    def struct(self) -> "_lib.HpmnScatterPlan":
        p = _lib.HpmnScatterPlan()
        p.n, p.perm, p.seg, p.start = self.n, self.perm.data_ptr(), self.seg.data_ptr(), self.start.data_ptr()
        p.rows, p.count, p.partials = self.rows.data_ptr(), self.count.data_ptr(), self.partials.data_ptr()
        p.out_rows = _ptr(self.out_rows)
        return p

    def record_stream(self, stream) -> None:
        """The plan was built on another stream than the one that consumes it."""
        for t in (self.sorted, self.workspace, self.perm, self.seg, self.start, self.rows, self.count, self.partials,
                  self.out_rows, self.counts_vec):
            if t is not None:
                t.record_stream(stream)


def set_scatter_plan(device, plan: Optional[ScatterPlan]) -> None:
    """hpmn_train_set_scatter_plan: the next hpmn_scan_bwd on the current stream's context scatters through ``plan``."""
    st = plan.struct() if plan is not None and plan.n > 0 else None
    _lib.check(_lib.load().hpmn_train_set_scatter_plan(_ctx(device), C.byref(st) if st is not None else None),
               "hpmn_train_set_scatter_plan")


def embed_grad_segsum(plan: ScatterPlan, ids_shape, d_x, d_emb, front_zero: int, mask_id0: bool, d_last=None, t_last: int = 0):
    """hpmn_embed_grad_segsum: the deterministic scatter as a call of its own (d_emb may be None when plan.out_rows is set)."""
    B, T, F = ids_shape
    _chk_f32(d_x, d_emb, d_last)
    st = plan.struct()
    rc = _lib.load().hpmn_embed_grad_segsum(C.byref(st), d_x.data_ptr(), _ptr(d_emb), B, T, F, plan.E, front_zero,
                                            plan.id_flags | (ID_MASK0 if mask_id0 else 0), _ptr(d_last), t_last, _stream())
    _lib.check(rc, "hpmn_embed_grad_segsum")


def train_probe(device, enable: bool) -> None:
    """hpmn_train_probe: bracket layer 0's reverse-scan launch of the following steps with timing events."""
    _lib.check(_lib.load().hpmn_train_probe(_ctx(device), int(enable)), "hpmn_train_probe")


def train_mark_layer0_reverse(device, enable: bool) -> None:
    """hpmn_train_mark_layer0_reverse: record an event in front of layer 0's reverse-scan launch of the following steps."""
    _lib.check(_lib.load().hpmn_train_mark_layer0_reverse(_ctx(device), int(enable)), "hpmn_train_mark_layer0_reverse")


def train_wait_layer0_reverse(device, stream) -> None:
    """hpmn_train_wait_layer0_reverse: ``stream`` waits for the start of layer 0's reverse-scan launch of the last
    hpmn_scan_bwd (no-op where that step had none)."""
    _lib.check(_lib.load().hpmn_train_wait_layer0_reverse(_ctx(device), stream.cuda_stream), "hpmn_train_wait_layer0_reverse")


def train_probe_ms(device) -> float:
    ms = C.c_float()
    _lib.check(_lib.load().hpmn_train_probe_ms(_ctx(device), C.byref(ms)), "hpmn_train_probe_ms")
    return float(ms.value)


class AbiSaved:
    """Saved states of hpmn_scan_fwd_train: one workspace + its layout.  Iterating yields the per-layer
    (x_in, hs, gates) views the per-layer Python path returns, for inspection."""

    def __init__(self, spec, B, V, workspace, layout, desc):
        self.spec, self.B, self.V, self.workspace, self.layout, self.desc = spec, B, V, workspace, layout, desc

    def _view(self, off, shape):
        base = (-self.workspace.data_ptr()) % 256
        n = 1
        for d_ in shape:
            n *= d_
        return self.workspace[base + off: base + off + 4 * n].view(torch.float32).view(*shape)

    def tensor(self, name, i=0):
        L, H, B = self.layout, self.spec.H, self.B
        T = L.T[i]
        D = self.spec.D0 if i == 0 else H
        if name == "x0":
            return self._view(L.x0, (B, L.T[0], self.spec.D0))
        shape = {"hs": (B, T + 1, H), "gates": (B, T, 3 * H), "y": (B, T // self.spec.periods[i], H),
                 "d_act": (B, T, 3 * H), "d_x": (B, T, D)}[name]
        return self._view(getattr(L, name)[i], shape)

    def __iter__(self):
        for i in range(self.spec.K):
            x_in = self.tensor("x0") if i == 0 else self.tensor("y", i - 1)
            yield (x_in, self.tensor("hs", i), self.tensor("gates", i))

    def __getitem__(self, i):
        return list(self)[i]


def _wptrs(weights, K, j):
    return (C.c_void_p * K)(*[weights[4 * i + j].data_ptr() for i in range(K)])


def abi_forward_train(spec: ScanSpec, ids, emb, weights: Sequence[torch.Tensor]):
    """hpmn_scan_fwd_train: the training forward of build_memory in ONE library call."""
    _chk_ids(ids)
    _chk_f32(emb, *weights)
    B, K, H = ids.shape[0], spec.K, spec.H
    V = emb.shape[0]
    lib = _lib.load()
    d = spec.desc(B, V, ids)
    lay = _lib.HpmnTrainLayout()
    _lib.check(lib.hpmn_scan_train_layout(C.byref(d), C.byref(lay)), "hpmn_scan_train_layout")
    ws = torch.empty(int(lay.total_bytes), device=emb.device, dtype=torch.uint8)
    memory = torch.empty(B, K, H, device=emb.device, dtype=torch.float32)
    last = torch.empty(B, spec.D0, device=emb.device, dtype=torch.float32)
    rc = lib.hpmn_scan_fwd_train(_ctx(emb.device), C.byref(d), ids.data_ptr(), emb.data_ptr(), _wptrs(weights, K, 0),
                                 _wptrs(weights, K, 1), _wptrs(weights, K, 2), _wptrs(weights, K, 3),
                                 memory.data_ptr(), last.data_ptr(), ws.data_ptr(), _stream())
    _lib.check(rc, "hpmn_scan_fwd_train")
    return memory, last, AbiSaved(spec, B, V, ws, lay, d)


class _AbiPending:
    def __init__(self, ctx, refs):
        self._ctx, self._refs = ctx, refs

    def join(self):
        if self._ctx is not None:
            _lib.check(_lib.load().hpmn_train_join(self._ctx, _stream()), "hpmn_train_join")
            self._ctx, self._refs = None, None


def abi_backward(spec: ScanSpec, ids, saved: "AbiSaved", weights, d_memory, d_last, grad_out, defer_join=False):
    """hpmn_scan_bwd: BPTT of abi_forward_train in ONE library call (weight gradients on the context's helper
    stream; with ``defer_join`` they may still be running when this returns -- see PendingGrads)."""
    K = spec.K
    _chk_f32(d_memory, d_last, *grad_out)
    assert d_memory.is_contiguous() and d_last.is_contiguous()
    gw = list(grad_out[1:])
    arr = lambda j: (C.c_void_p * K)(*[gw[4 * i + j].data_ptr() for i in range(K)])
    ctx = _ctx(d_memory.device)
    saved.desc.mask_id0 = spec.id_flags(ids)           # (the scatter's ids may be narrower than the forward's: lazy table Adam)
    rc = _lib.load().hpmn_scan_bwd(ctx, C.byref(saved.desc), ids.data_ptr(), _wptrs(weights, K, 0),
                                   _wptrs(weights, K, 2), d_memory.data_ptr(), d_last.data_ptr(), arr(0), arr(1),
                                   arr(2), arr(3), _ptr(grad_out[0]), saved.workspace.data_ptr(),
                                   int(defer_join), _stream())
    _lib.check(rc, "hpmn_scan_bwd")
    if defer_join:
        return _AbiPending(ctx, (saved, weights))
    return None


def scan_forward_train(spec: ScanSpec, ids, emb, weights: Sequence[torch.Tensor]):
    """Training-mode build_memory.  Default: the whole graph in one library call (hpmn_scan_fwd_train); the
    pipelined launches (pipe_mode) and the per-layer Python orchestration below are options."""
    mode = pipe_mode(spec) if ids.shape[0] > 0 else ""
    if mode == "all":
        return pipe_forward(spec, ids, emb, weights, train=True)
    if mode == "upper":
        return upper_forward(spec, ids, emb, weights)
    if TRAIN_ABI and PIPELINE_CHUNKS <= 1 and FUSED_FWD and not SPLIT_LAYER0_BWD and ids.shape[0] > 0:
        return abi_forward_train(spec, ids, emb, weights)
    return scan_forward_train_layers(spec, ids, emb, weights)


def scan_forward_train_layers(spec: ScanSpec, ids, emb, weights: Sequence[torch.Tensor]):
    """Training-mode build_memory (code/hpmn.py:113-129 on the embedded ids): per layer the input
    projection + the serial scan with saved states.  Returns (memory [B,K,H], last [B,D0], saved).

    Layer i+1 only needs every period-th output of layer i, so the layers are software-pipelined in time
    chunks over K HIP streams: while layer 0 scans chunk c+1, layer 1 projects+scans chunk c, layer 2
    chunk c-1, ...  The serial chain shrinks from sum_i T_i steps to about T_0 + T_1/nc + T_2/nc + ...
    (at B ~ 2 sequences per CU a scan launch occupies one wave on half of the SIMDs)."""
    lens = spec.layer_lengths()
    B = ids.shape[0]
    H, K = spec.H, spec.K
    dev = emb.device
    f32 = dict(device=dev, dtype=torch.float32)
    memory = torch.empty(B, K, H, **f32)
    in_dims = [spec.D0] + [H] * (K - 1)
    x0 = torch.empty(B, lens[0], spec.D0, **f32)
    # the fused layer spends a second wave per sequence on a SIMD that would otherwise idle: a win while
    # 2 B waves still find (about) a SIMD each -- measured at C3: +3.7 % at B=500, -6.6 % at B=750 / 1000
    room = 2 * B <= 1.1 * 4 * torch.cuda.get_device_properties(dev).multi_processor_count
    fused = [room and PIPELINE_CHUNKS <= 1 and fused_fwd_supported(H, in_dims[i], i == 0) and
             (i > 0 or 64 % spec.E == 0) for i in range(K)]
    xp = [None if fused[i] else torch.empty(B, lens[i], 3 * H, **f32) for i in range(K)]
    hs = [torch.empty(B, lens[i] + 1, H, **f32) for i in range(K)]
    gates = [torch.empty(B, lens[i], 3 * H, **f32) for i in range(K)]
    y = [torch.empty(B, lens[i] // spec.periods[i], H, **f32) if i + 1 < K else None for i in range(K)]
    x_in = [x0] + y[:-1]

    plan = chunk_plan(spec, PIPELINE_CHUNKS)
    nc = PIPELINE_CHUNKS if plan is not None else 1
    main = torch.cuda.current_stream()
    streams = [main] + (_streams(dev, K - 1) if nc > 1 else [main] * (K - 1))
    if nc > 1:
        for st in streams[1:]:
            st.wait_stream(main)
    done = [[None] * nc for _ in range(K)]
    for c in range(nc):
        for i in range(K):
            wg, bg, wc, bc = weights[4 * i:4 * i + 4]
            L = plan[i] if plan is not None else lens[i]
            t0, t1 = c * L, (c + 1) * L
            with torch.cuda.stream(streams[i]):
                if i > 0 and nc > 1:
                    streams[i].wait_event(done[i - 1][c])
                if fused[i] and nc == 1:
                    if i == 0:
                        gru_fused_fwd(ids=ids, emb=emb, wg=wg, bg=bg, wc=wc, bc=bc, H=H, T=lens[0],
                                      front_zero=spec.front_zero, mask_id0=spec.mask_id0, h_last=memory[:, 0, :],
                                      period=spec.periods[0], out=(y[0], hs[0], gates[0], x0))
                    else:
                        gru_fused_fwd(x=x_in[i], wg=wg, bg=bg, wc=wc, bc=bc, H=H, T=lens[i], h_last=memory[:, i, :],
                                      period=spec.periods[i], out=(y[i], hs[i], gates[i], None))
                    continue
                if i == 0:
                    gru_input_proj(None, ids=ids, emb=emb, wg=wg, bg=bg, wc=wc, bc=bc, H=H, T=lens[0],
                                   front_zero=spec.front_zero, mask_id0=spec.mask_id0, out=(xp[0], x0),
                                   t_range=(t0, L) if nc > 1 else None)
                else:
                    gru_input_proj(None, x=x_in[i], wg=wg, bg=bg, wc=wc, bc=bc, H=H, T=lens[i], out=(xp[i], None),
                                   t_range=(t0, L) if nc > 1 else None)
                gru_scan_fwd(xp[i], wg, wc, in_dims[i], memory[:, i, :], spec.periods[i], want_y=(i + 1 < K),
                             train=True, out=(y[i], hs[i], gates[i]),
                             t_range=(t0, t1) if nc > 1 else None,
                             h_init=hs[i][:, t0, :] if (nc > 1 and c > 0) else None)
                if nc > 1:
                    done[i][c] = torch.cuda.Event()
                    done[i][c].record(streams[i])
    if nc > 1:
        for st in streams[1:]:
            main.wait_stream(st)
    last = x0[:, spec.last_index, :].contiguous()
    saved = [(x_in[i], hs[i], gates[i]) for i in range(K)]
    del xp            # only now: the side streams are joined
    return memory, last, saved


# measured neutral (C3 4.143 vs 4.169 ms/step, C4 10.48 vs 10.33): off by default, kept as an option
SPLIT_LAYER0_BWD = int(os.environ.get("HPMN_SPLIT_LAYER0_BWD", "0")) != 0


def _time_cut(T: int, period: int) -> int:
    """Step at which a long reverse scan is cut in two launches (a multiple of 2 and of the period, about the
    middle); 0 = do not cut (short sequences: the extra launch costs more than it hides)."""
    if T < 256:
        return 0
    q = 2 * period // math.gcd(2, period)
    cut = (T // 2) // q * q
    return cut if 0 < cut < T else 0


class PendingGrads:
    """Weight-gradient work still running on a side stream when scan_backward(defer_join=True) returns:
    the embedding gradient is complete on the current stream, the GRU weight gradients are not until
    ``join()``.  Holds the buffers that work reads so the caching allocator cannot recycle them early."""

    def __init__(self, streams, refs):
        self._streams, self._refs = list(streams), refs

    def join(self):
        main = torch.cuda.current_stream()
        for st in self._streams:
            main.wait_stream(st)
        self._streams, self._refs = [], None


def scan_backward(spec: ScanSpec, ids, saved, weights: Sequence[torch.Tensor], d_memory, d_last, grad_out,
                  defer_join: bool = False, scatter_plan: Optional["ScatterPlan"] = None):
    """BPTT of scan_forward_train (every forward path leaves the same saved states).  ``scatter_plan``: the library step
    scatters through it (deterministic segmented reduction); the per-layer measurement paths keep the atomic kernel."""
    if isinstance(saved, AbiSaved):
        set_scatter_plan(d_memory.device, scatter_plan)
        return abi_backward(spec, ids, saved, weights, d_memory, d_last, grad_out, defer_join)
    mode = pipe_mode(spec) if d_memory.shape[0] > 0 else ""
    if mode in ("all", "upper"):
        if scatter_plan is not None:
            #  these launches scatter atomically on their own: a plan would be ignored and its compact rows
            # -- what the data-parallel exchange sends -- never written
            raise RuntimeError("HPMN_PIPE=%s cannot scatter through a ScatterPlan (set HPMN_DET_SCATTER=0 / HPMN_ROWS_TAIL=0)" % mode)
        fn = pipe_backward if mode == "all" else upper_backward
        return fn(spec, ids, saved, weights, d_memory, d_last, grad_out, defer_join)
    return scan_backward_layers(spec, ids, saved, weights, d_memory, d_last, grad_out, defer_join, scatter_plan=scatter_plan)


def _scatter(plan, ids, d_x0, d_emb, spec):
    """The per-layer paths' scatter: through the plan when one is given (its compact rows are what the caller consumes)."""
    if plan is not None and plan.n > 0:
        embed_grad_segsum(plan, tuple(ids.shape), d_x0, d_emb, spec.front_zero, spec.mask_id0)
    else:
        embed_grad_scatter(ids, d_x0, d_emb, spec.front_zero, spec.mask_id0)


def scan_backward_layers(spec: ScanSpec, ids, saved, weights: Sequence[torch.Tensor], d_memory, d_last, grad_out,
                         defer_join: bool = False, scatter_plan: Optional["ScatterPlan"] = None):
    """BPTT of scan_forward_train_layers.  ``grad_out`` = [d_emb, d_wg0, d_bg0, d_wc0, d_bc0, d_wg1, ...]:
    pre-zeroed buffers (views of the optimiser's flat gradient) that are accumulated into.

    Mirror image of the forward pipeline: layer K-1 runs its last time chunk first, its input gradient
    feeds layer K-2's last chunk, ... and layer 0 works on chunk c while the layers above are already on
    chunk c-1.  The weight gradient of a layer (an MFMA reduction over the whole d_act) is off the serial
    chain and is issued on the layer's stream once its last chunk is done."""
    K, H = spec.K, spec.H
    lens = spec.layer_lengths()
    d_emb, gw = grad_out[0], list(grad_out[1:])
    dev = d_memory.device
    f32 = dict(device=dev, dtype=torch.float32)
    B = d_memory.shape[0]
    in_dims = [spec.D0] + [H] * (K - 1)
    d_act = [torch.empty(B, lens[i], 3 * H, **f32) for i in range(K)]
    d_x = [torch.empty(B, lens[i], in_dims[i], **f32) for i in range(K)]      # d_x[i+1] is layer i's d_y
    carry = [torch.empty(B, H, **f32) for _ in range(K)]
    plan = chunk_plan(spec, PIPELINE_CHUNKS)
    nc = PIPELINE_CHUNKS if plan is not None else 1
    main = torch.cuda.current_stream()
    keep = []
    if nc == 1:
        # unpipelined: the whole chain on the main stream, weight gradients underneath on one side stream
        side = _streams(dev, 1)[0]
        for i in range(K - 1, -1, -1):
            wg, bg, wc, bc = weights[4 * i:4 * i + 4]
            x_in, hs, gates = saved[i]
            d_y = d_x[i + 1] if i + 1 < K else None
            # Layer 0 is the end of the chain: nothing runs under ITS weight-gradient reduction except the
            # scatter, so the reverse scan is cut in two time halves and the reduction over the late half starts
            # (on the side stream) while the scan works on the early half.
            cut = _time_cut(lens[i], spec.periods[i]) if (i == 0 and SPLIT_LAYER0_BWD) else 0
            if cut:
                gru_scan_bwd(wg, wc, in_dims[i], hs, gates, d_memory[:, i, :], d_y, spec.periods[i], out=d_act[i],
                             t_range=(cut, lens[i]), dh_carry=carry[i])
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    gru_param_grads(x_in, hs, gates, d_act[i], wg, wc, gw[4 * i], gw[4 * i + 1], gw[4 * i + 2],
                                    gw[4 * i + 3], want_dx=False, keep=keep, t_range=(cut, lens[i] - cut))
                gru_scan_bwd(wg, wc, in_dims[i], hs, gates, d_memory[:, i, :], d_y, spec.periods[i], out=d_act[i],
                             t_range=(0, cut), dh_carry=carry[i])
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    gru_param_grads(x_in, hs, gates, d_act[i], wg, wc, gw[4 * i], gw[4 * i + 1], gw[4 * i + 2],
                                    gw[4 * i + 3], want_dx=False, keep=keep, t_range=(0, cut))
            else:
                fdx = scan_bwd_fuses_dx(H, B) and in_dims[i] in (16, 32, 64)
                gru_scan_bwd(wg, wc, in_dims[i], hs, gates, d_memory[:, i, :], d_y, spec.periods[i], out=d_act[i],
                             d_x=d_x[i] if fdx else None)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    gru_param_grads(x_in, hs, gates, d_act[i], wg, wc, gw[4 * i], gw[4 * i + 1], gw[4 * i + 2],
                                    gw[4 * i + 3], want_dx=False, keep=keep, whole_cu=(i == 0 and H <= 64))
            if cut or not (scan_bwd_fuses_dx(H, B) and in_dims[i] in (16, 32, 64)):
                gru_input_grad(d_act[i], wg, wc, in_dims[i], out=d_x[i])
        d_x0 = d_x[0]
        d_x0[:, spec.last_index, :] += d_last
        _scatter(scatter_plan, ids, d_x0, d_emb, spec)
        pending = PendingGrads([side], (keep, d_act, d_x, saved))
        if defer_join:
            return pending
        pending.join()
        return None
    # layer 0 (the end of the chain, feeding the embedding scatter) stays on the main stream
    streams = [main] + _streams(dev, K - 1)
    for st in streams[1:]:
        st.wait_stream(main)
    done = [[None] * nc for _ in range(K)]
    for c in range(nc - 1, -1, -1):
        for i in range(K - 1, -1, -1):
            wg, bg, wc, bc = weights[4 * i:4 * i + 4]
            x_in, hs, gates = saved[i]
            L = plan[i] if plan is not None else lens[i]
            t0, t1 = c * L, (c + 1) * L
            with torch.cuda.stream(streams[i]):
                if i + 1 < K:
                    streams[i].wait_event(done[i + 1][c])
                gru_scan_bwd(wg, wc, in_dims[i], hs, gates, d_memory[:, i, :], d_x[i + 1] if i + 1 < K else None,
                             spec.periods[i], out=d_act[i], t_range=(t0, t1) if nc > 1 else None,
                             dh_carry=carry[i] if nc > 1 else None)
                gru_input_grad(d_act[i], wg, wc, in_dims[i], out=d_x[i], t_range=(t0, L) if nc > 1 else None)
                done[i][c] = torch.cuda.Event()
                done[i][c].record(streams[i])
                if c == 0 and i > 0:        # this layer is finished: its weight gradient, off the chain
                    gru_param_grads(x_in, hs, gates, d_act[i], wg, wc, gw[4 * i], gw[4 * i + 1], gw[4 * i + 2],
                                    gw[4 * i + 3], want_dx=False, keep=keep)
    # layer 0's weight gradient overlaps the scatter on a side stream
    side = streams[1] if K > 1 else _streams(dev, 1)[0]
    side.wait_event(done[0][0])
    with torch.cuda.stream(side):
        wg, bg, wc, bc = weights[0:4]
        gru_param_grads(saved[0][0], saved[0][1], saved[0][2], d_act[0], wg, wc, gw[0], gw[1], gw[2], gw[3],
                        want_dx=False, keep=keep)
    d_x0 = d_x[0]
    d_x0[:, spec.last_index, :] += d_last
    _scatter(scatter_plan, ids, d_x0, d_emb, spec)
    pending = PendingGrads(set(streams[1:] + [side]), (keep, d_act, d_x, saved))
    if defer_join:
        return pending
    pending.join()
    return None


# ---------------------------------------------------------------------------------------
# read path (covariance regulariser + attention hops + head + loss)
# ---------------------------------------------------------------------------------------
_read_ws = {}
_read_ws_need = {}


def _read_workspace(descs, device):
    """Zero-initialised once, then reused: [16 slabs of the parameter range (only parameter positions are ever rewritten) |
    the tape of the weight-gradient products (rewritten whole by every training launch)]."""
    descs = list(descs) if isinstance(descs, (list, tuple)) else [descs]
    desc = descs[0]
    nkey = (int(desc.B), int(desc.n_params)) + tuple((int(d.K), int(d.H), int(d.D0), int(d.hop)) for d in descs)
    need = _read_ws_need.get(nkey)
    if need is None:                # (asked once per shape: the call builds ctypes pointer objects, garbage for the cycle collector)
        need = _read_ws_need[nkey] = _lib.load().hpmn_read_workspace_bytes_n(len(descs), _desc_array(descs)) // 4
    key = (str(device), int(desc.n_params), torch.cuda.current_stream().cuda_stream)
    ws = _read_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _read_ws[key] = torch.zeros(need, device=device, dtype=torch.float32)
    return ws


def read_fwd(desc, params, memory, last, want_logit=False, want_att=False):
    """hpmn_read_fwd -> dict(prediction, logit?, user_weights?, memory_loss)."""
    _chk_f32(params, memory, last)
    B, K, H = memory.shape
    desc.B = B
    dev = memory.device
    pred = torch.empty(B, device=dev)
    logit = torch.empty(B, device=dev) if want_logit else None
    att = torch.empty(B, K, device=dev) if want_att else None
    mem_loss = torch.zeros(1, device=dev)
    rc = _lib.load().hpmn_read_fwd(C.byref(desc), params.data_ptr(), memory.data_ptr(), last.data_ptr(),
                                    pred.data_ptr(), _ptr(logit), _ptr(att), mem_loss.data_ptr(), _stream())
    _lib.check(rc, "hpmn_read_fwd")
    return dict(prediction=pred, logit=logit, user_weights=att, memory_loss=mem_loss[0], memory=memory)


def read_fwd_bwd(desc, params, d_params, memory, last, label, masks, keep_prob, inv_global_batch, memory_reg,
                 dropout_seed: int = 0, loss_out: Optional[torch.Tensor] = None, defer_param_grads: bool = False):
    """hpmn_read_fwd_bwd: forward + loss + backward of the read path; accumulates into d_params.
    ``masks`` = (mask1 [B,200], mask2 [B,80]) or None; with None and keep_prob < 1 a non-zero ``dropout_seed``
    makes the kernel draw the masks itself.  ``loss_out``: a ZEROED [2] buffer to accumulate the two loss sums into
    (the training step keeps one and clears it off the critical path); default: a fresh one.
    ``defer_param_grads``: leave the parameter gradients as partial sums in the workspace; the returned dict then holds
    ``reduce_param_grads``, a callable that adds them to ``d_params`` on whatever stream is current when it is called."""
    _chk_f32(params, d_params, memory, last)
    B, K, H = memory.shape
    desc.B = B
    desc.dropout_seed = int(dropout_seed) & 0xFFFFFFFFFFFFFFFF
    dev = memory.device
    assert label.dtype == torch.int32 and label.is_contiguous()
    pred = torch.empty(B, device=dev)
    if loss_out is None:
        loss_out = torch.zeros(2, device=dev)
    d_memory = torch.empty_like(memory)
    d_last = torch.empty_like(last)
    m1 = m2 = None
    if masks is not None:
        m1, m2 = masks
        _chk_f32(m1, m2)
    ws = _read_workspace(desc, dev)
    rc = _lib.load().hpmn_read_fwd_bwd(C.byref(desc), params.data_ptr(), memory.data_ptr(), last.data_ptr(),
                                        label.data_ptr(), _ptr(m1), _ptr(m2), float(keep_prob),
                                        float(inv_global_batch), float(memory_reg), pred.data_ptr(),
                                        loss_out.data_ptr(), d_memory.data_ptr(), d_last.data_ptr(),
                                        None if defer_param_grads else d_params.data_ptr(), ws.data_ptr(), _stream())
    _lib.check(rc, "hpmn_read_fwd_bwd")
    out = dict(prediction=pred, log_loss_sum=loss_out[0], memory_loss=loss_out[1], d_memory=d_memory, d_last=d_last)
    if defer_param_grads:
        def reduce_param_grads(loss3: Optional[torch.Tensor] = None):
            """adds the read path's weight gradients to d_params (current stream); with ``loss3`` (a [3] buffer) the same
            launches also leave {log-loss sum, memory-loss sum, cross_entropy} there and clear ``loss_out``."""
            if loss3 is None:
                _lib.check(_lib.load().hpmn_read_param_grads(C.byref(desc), d_params.data_ptr(), ws.data_ptr(), _stream()),
                           "hpmn_read_param_grads")
            else:
                _lib.check(_lib.load().hpmn_read_param_grads_loss_n(1, _desc_array([desc]), d_params.data_ptr(), ws.data_ptr(),
                                                                     loss_out.data_ptr(), float(inv_global_batch),
                                                                     float(memory_reg), loss3.data_ptr(), _stream()),
                           "hpmn_read_param_grads_loss_n")
        out["reduce_param_grads"] = reduce_param_grads
    return out


def _desc_array(descs):
    arr = (C.POINTER(_lib.HpmnReadDesc) * len(descs))()
    for i, d in enumerate(descs):
        arr[i] = C.pointer(d)
    return arr


def _ptr_array(ts):
    return (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def read_fwd_n(descs, params, memories, lasts, want_logit=True, want_att=True):
    """hpmn_read_fwd_n: the read path of a graph with ``len(descs)`` branches (user, item) in one launch
    -> dict(prediction, logit?, weights = [first-hop attention weights per branch], memory_loss)."""
    _chk_f32(params, *memories, *lasts)
    B = memories[0].shape[0]
    dev = params.device
    for d in descs:
        d.B = B
    pred = torch.empty(B, device=dev)
    logit = torch.empty(B, device=dev) if want_logit else None
    atts = [torch.empty(B, m.shape[1], device=dev) if want_att else None for m in memories]
    mem_loss = torch.zeros(1, device=dev)
    rc = _lib.load().hpmn_read_fwd_n(len(descs), _desc_array(descs), params.data_ptr(), _ptr_array(memories),
                                      _ptr_array(lasts), pred.data_ptr(), _ptr(logit), _ptr_array(atts),
                                      mem_loss.data_ptr(), _stream())
    _lib.check(rc, "hpmn_read_fwd_n")
    return dict(prediction=pred, logit=logit, weights=atts, memory_loss=mem_loss[0])


def read_fwd_bwd_n(descs, params, d_params, memories, lasts, label, masks, keep_prob, inv_global_batch, memory_reg,
                   dropout_seed: int = 0, loss_out: Optional[torch.Tensor] = None):
    """hpmn_read_fwd_bwd_n: forward + loss + backward of the read path of a graph with several branches; accumulates into
    d_params.  -> dict(prediction, log_loss_sum, memory_loss, d_memory = [...], d_last = [...])."""
    _chk_f32(params, d_params, *memories, *lasts)
    B = memories[0].shape[0]
    dev = params.device
    for d in descs:
        d.B = B
    descs[0].dropout_seed = int(dropout_seed) & 0xFFFFFFFFFFFFFFFF
    assert label.dtype == torch.int32 and label.is_contiguous()
    pred = torch.empty(B, device=dev)
    if loss_out is None:
        loss_out = torch.zeros(2, device=dev)
    d_mem = [torch.empty_like(m) for m in memories]
    d_last = [torch.empty_like(l) for l in lasts]
    m1 = m2 = None
    if masks is not None:
        m1, m2 = masks
        _chk_f32(m1, m2)
    ws = _read_workspace(descs, dev)
    rc = _lib.load().hpmn_read_fwd_bwd_n(len(descs), _desc_array(descs), params.data_ptr(), _ptr_array(memories),
                                          _ptr_array(lasts), label.data_ptr(), _ptr(m1), _ptr(m2), float(keep_prob),
                                          float(inv_global_batch), float(memory_reg), pred.data_ptr(), loss_out.data_ptr(),
                                          _ptr_array(d_mem), _ptr_array(d_last), d_params.data_ptr(), ws.data_ptr(), _stream())
    _lib.check(rc, "hpmn_read_fwd_bwd_n")
    return dict(prediction=pred, log_loss_sum=loss_out[0], memory_loss=loss_out[1], d_memory=d_mem, d_last=d_last)
