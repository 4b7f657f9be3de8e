"""Online (serving-time) form of the periodic memory: one new event per user and call against a persisted
per-user state store, instead of replaying the user's lifelong sequence.

Reference: ``Memory_Ind`` of code/srnn.py:699-796 -- a ``[users, 9*H]`` state variable, gathered by
``user_id`` (:790-792), pushed through the cascade "layer i fires iff index % 2^i == 0" (:727-748) and
scattered back (:794-796) -- here with the periods of code/hpmn.py:113-129 and the weights of a trained
``Hpmn`` / ``Hpmn_Industry`` model.  After T events the store holds the ``memory`` the batch scan computes
over those T steps (tests/test_gpu_parity.py), so prediction for a stored user is one read-path launch.

What "the same as the trained graph" means here:
  * ``Hpmn_Industry`` scans 23 all-zero steps in front of every sequence (code/hpmn.py:288-289).  They are not
    no-ops (gate bias 1, no masking) and they shift the layer-firing phase by an odd offset, so a NEW user starts
    from the state and event count those ``spec.front_zero`` steps leave (computed once, with the same kernel).
  * the query row is ``uinp[:, last_index, :]``: the candidate itself for ``Hpmn`` (-1, code/hpmn.py:439), the
    event BEFORE the candidate for ``Hpmn_Industry`` (-2, code/hpmn.py:292) -- in both graphs the candidate is
    also the last step of the scan.  ``predict`` therefore scores "the user's last stored event is the
    candidate" and takes the query row from the store (``last_x`` / ``prev_x``).
  * front PADDING of ``dataset_hpmn.pkl`` samples is a batch artefact (pad steps are real zero-input steps for the
    batch scan); an online store equals the batch scan over exactly the events it was fed, i.e. a sample with no
    padding.
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
        self.last_index = int(spec.last_index)
        if self.last_index not in (-1, -2):
            raise ValueError("last_index %d: only the reference's -1 / -2 query rows are supported" % self.last_index)
        self.state = torch.zeros(n_users, self.K, self.H, device=dev, dtype=torch.float32)
        self.count = torch.zeros(n_users, device=dev, dtype=torch.int32)
        # input rows of the most recent event and of the one before it (query row for last_index -1 / -2)
        self.last_x = torch.zeros(n_users, self.D0, device=dev, dtype=torch.float32)
        self.prev_x = torch.zeros(n_users, self.D0, device=dev, dtype=torch.float32)
        self.front_zero = int(spec.front_zero)
        if self.front_zero and n_users:
            # every new user starts from the state after the graph's all-zero prefix: run it once on user 0,
            # then broadcast (state, count) to the whole store
            u0 = torch.zeros(1, device=dev, dtype=torch.int32)
            z = torch.zeros(1, self.D0, device=dev, dtype=torch.float32)
            for _ in range(self.front_zero):
                self._update_kernel(u0, z)
            self.state[:] = self.state[0].clone()
            self.count[:] = self.front_zero

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
        self._update_kernel(user, x)
        u = user.long()
        self.prev_x[u] = self.last_x[u]
        self.last_x[u] = x

    def events(self, user: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Real events fed so far (the all-zero prefix of the graph is not counted)."""
        c = self.count if user is None else self.count[user.long()]
        return c - self.front_zero

    def _update_kernel(self, user: torch.Tensor, x: torch.Tensor) -> None:
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

    def predict(self, user: torch.Tensor, target_ids: Optional[torch.Tensor] = None):
        """Score stored users whose LAST fed event is the candidate (both reference graphs scan the candidate as
        the final step): the model's read path (query_memory + head, code/hpmn.py:172-199) over the stored
        memory with ``last`` = the graph's query row -- the candidate's own input row for ``Hpmn``
        (``last_index`` -1; re-embedded from ``target_ids`` when given, which must then be the ids of that last
        event), the event before it for ``Hpmn_Industry`` (``last_index`` -2)."""
        m = self.model
        B = user.shape[0]
        if self.last_index == -2:
            last = self.prev_x[user.long()]
        elif target_ids is not None:
            last = ops.embed_gather(target_ids.reshape(B, 1, -1).contiguous(), m.params["Embedding/emb_mtx"],
                                    m.spec.mask_id0).reshape(B, -1)
        else:
            last = self.last_x[user.long()]
        return ops.read_fwd(m._read_desc, m._read_params, self.memory(user).contiguous(), last.contiguous(),
                            want_logit=True)
