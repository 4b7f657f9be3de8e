"""Online (serving-time) form of the periodic memory: one new event per user and call against a persisted
per-user state store, instead of replaying the user's lifelong sequence.

Reference: ``Memory_Ind`` of code/srnn.py:699-796 -- a ``[users, 9*H]`` state variable, gathered by
``user_id`` (:790-792), pushed through the cascade "layer i fires iff index % 2^i == 0" (:727-748) and
scattered back (:794-796) -- here with the periods of code/hpmn.py:113-129 and the weights of a trained
``Hpmn`` / ``Hpmn_Industry`` model.  After T events the store holds the ``memory`` the batch scan computes
over those T steps (tests/test_gpu_parity.py), so prediction for a stored user is one read-path launch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, ops


class OnlineMemory:
    """State store ``[n_users, K, H]`` + per-user event counters on the model's device."""

    def __init__(self, model, n_users: int):
        self.model = model
        spec = model.spec
        self.K, self.H, self.D0 = spec.K, spec.H, spec.D0
        self.periods = tuple(spec.periods[:spec.K])
        dev = model.device
        self.state = torch.zeros(n_users, self.K, self.H, device=dev, dtype=torch.float32)
        self.count = torch.zeros(n_users, device=dev, dtype=torch.int32)

    def update(self, user: torch.Tensor, ids: torch.Tensor) -> None:
        """One event per listed user: ``user`` [B] int32 rows of the store (distinct), ``ids`` [B, F] the event's
        feature ids (embedded with the model's table, id-0 mask as in the model's graph)."""
        m = self.model
        assert user.dtype == torch.int32 and user.is_cuda and user.is_contiguous() and user.dim() == 1
        B = user.shape[0]
        if B == 0:
            return
        x = ops.embed_gather(ids.reshape(B, 1, -1).contiguous(), m.params["Embedding/emb_mtx"], m.spec.mask_id0)
        self.update_rows(user, x.reshape(B, -1))

    def update_rows(self, user: torch.Tensor, x: torch.Tensor) -> None:
        """Same with the input rows ``x`` [B, D0] already formed."""
        a = _lib.HpmnOnlineUpdate()
        B = user.shape[0]
        a.B, a.D, a.H, a.K = B, self.D0, self.H, self.K
        for i, p in enumerate(self.periods):
            a.periods[i] = int(p)
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and tuple(x.shape) == (B, self.D0)
        a.user, a.x, a.state, a.count = user.data_ptr(), x.data_ptr(), self.state.data_ptr(), self.count.data_ptr()
        w = self.model._gru_weights()
        for i in range(self.K):
            a.wg[i], a.bg[i], a.wc[i], a.bc[i] = (t.data_ptr() for t in w[4 * i:4 * i + 4])
        rc = _lib.load().hpmn_memory_update(C.byref(a), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "hpmn_memory_update")

    def memory(self, user: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[B, K, H] memory of the listed users (all users when None)."""
        return self.state if user is None else self.state[user.long()]

    def predict(self, user: torch.Tensor, target_ids: torch.Tensor):
        """Score a candidate event for stored users: the model's read path (query_memory + head,
        code/hpmn.py:172-199) over the stored memory with ``last`` = the candidate's input row."""
        m = self.model
        B = user.shape[0]
        last = ops.embed_gather(target_ids.reshape(B, 1, -1).contiguous(), m.params["Embedding/emb_mtx"],
                                m.spec.mask_id0).reshape(B, -1)
        return ops.read_fwd(m._read_desc, m._read_params, self.memory(user).contiguous(), last.contiguous(),
                            want_logit=True)
