"""HPMN train/eval entrypoint on MI355X -- drop-in for the reference's ``code/hpmn.py``.

Same classes (``Hpmn_Basic`` / ``Hpmn_Industry`` / ``Hpmn``), constructor arguments,
``train/eval/save_model/load_model/get_weights/log`` methods, ``result.log`` line format and
CLI (``python hpmn.py <amazon|taobao|xlong>``) as /root/reference/code/hpmn.py; everything
below ``sess.run`` is replaced: the embedding gather, the periodic GRU memory update (forward
and BPTT), the embedding-gradient scatter and the TF-form Adam run in ``libhpmn_hip.so``
(hand-written HIP for gfx950), as does the memory read path (attention hops, head, loss, and
their backward); PyTorch-ROCm supplies device memory, streams, the dropout RNG and RCCL (``torch.distributed`` backend ``nccl``) for data parallel.

There is no CPU fallback: constructing a model without a GPU + the built library raises.
"""
from __future__ import annotations

import math
import os
import sys
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import dist, ops
from .data_loader import DataLoader, DataLoader_Mul, load_xlong_tsv
from .ops import ScanSpec

BN_EPS = 1e-3        # tf.layers.batch_normalization default (code/hpmn.py:190)
LOGLOSS_EPS = 1e-7   # tf.losses.log_loss default (code/hpmn.py:202)


def _glorot_uniform_(t: torch.Tensor, gen: torch.Generator):
    fan_in, fan_out = t.shape[0], t.shape[1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    t.uniform_(-lim, lim, generator=gen)


def _splitmix64(x: int) -> int:
    """One splitmix64 step: a well-mixed 64-bit value per counter value.  The per-step dropout seed must not be
    an arithmetic progression: the kernel adds sample/unit offsets to it, and with a linear seed the mask of
    (step, sample) would recur as that of (step-1, sample+4) -- always on a sample of the same label parity in
    alternating pos/neg data, which is a label leak the head learns within 50 steps."""
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def device_auc(pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """ROC AUC = (sum of the positives' mid-ranks - n1 (n1 + 1) / 2) / (n1 n0), ties sharing their average
    rank -- the Mann-Whitney form of the trapezoidal area sklearn.metrics.roc_auc_score integrates."""
    pred = pred.double()
    y = label.to(pred.device).double()
    order = torch.argsort(pred)
    ps = pred[order]
    n = ps.numel()
    # groups of equal predictions: every member gets the mean of the group's 1-based positions = (first + last) / 2 + 1.
    # First / last position of an element's group by two scans (cummax of the group heads, reversed cummin of the group tails)
    # -- no index_add_: with heavy ties (a collapsed model predicts ONE value: r4, bench.py's random-label rows after 200
    # steps) 16 000 double-precision atomics on one address took 7 ms each way and eval() 100 ms instead of 12.
    idx = torch.arange(n, device=pred.device)
    head = torch.ones(n, dtype=torch.bool, device=pred.device)
    head[1:] = ps[1:] != ps[:-1]
    tail = torch.ones(n, dtype=torch.bool, device=pred.device)
    tail[:-1] = head[1:]
    first = torch.cummax(torch.where(head, idx, torch.zeros_like(idx)), 0).values
    last = torch.flip(torch.cummin(torch.flip(torch.where(tail, idx, torch.full_like(idx, n)), [0]), 0).values, [0])
    rank = (first + last).double() / 2.0 + 1.0
    ys = y[order]
    n1 = ys.sum()
    n0 = n - n1
    # (one class only: n1 * n0 == 0 -> NaN here; eval() turns that into sklearn's ValueError on the host)
    return ((rank * ys).sum() - n1 * (n1 + 1) / 2) / (n1 * n0)


def device_log_loss(pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """sklearn.metrics.log_loss for binary labels: mean negative log-likelihood with the predictions
    clipped to [eps, 1 - eps], eps = float64 machine epsilon."""
    p = pred.double()
    y = label.to(p.device).double()
    eps = 2.220446049250313e-16
    p = p.clamp(eps, 1.0 - eps)
    return -(y * torch.log(p) + (1.0 - y) * torch.log1p(-p)).mean()


class _DeviceDataset:
    """int32 device-resident copy of a dataset (list-of-samples or XLong TSV path), built once.
    Replaces the per-batch Python-list -> ndarray -> feed_dict conversion of
    code/data_loader.py:283-296 / code/hpmn.py:474-481; batches are slices in stored order,
    exactly the batches ``DataLoader`` would yield."""

    def __init__(self, dataset, device, industry: bool, feature_size: Optional[int] = None, want_item: bool = False):
        item_ids = None
        if isinstance(dataset, dict):
            ids, label = dataset["ids"], dataset["label"]
            length = dataset.get("length")
            item_ids = dataset.get("item_ids")
        elif isinstance(dataset, str):
            # XLong TSV (code/data_loader.py:56-85): parsed once by a process pool into an array cache next to the file
            z = load_xlong_tsv(dataset)
            ids, label = z["ids"], z["label"]
            item_ids = z["item_ids"] if want_item else None     # item_inp <- data[3] (code/hpmn.py:330)
            length = None
        else:
            label = np.asarray([s[0] for s in dataset], dtype=np.int32)
            ids = np.asarray([s[1] for s in dataset], dtype=np.int32)
            length = np.asarray([s[2] for s in dataset], dtype=np.int32)
            if want_item:
                item_ids = np.asarray([s[3] for s in dataset], dtype=np.int32)   # item_inp <- data[3] (:477)
        if want_item and item_ids is None:
            raise ValueError("item=True needs the item-side sequences (sample[3] / dataset['item_ids'])")
        self.n = int(ids.shape[0])
        # tf.nn.embedding_lookup on the CPU raises on an out-of-range id; the kernels index the table unchecked,
        # so the range is checked once here, on the host, when the dataset is staged
        if feature_size is not None and self.n:
            lo, hi = int(np.min(ids)), int(np.max(ids))
            if lo < 0 or hi >= feature_size:
                # HPMN_OOB_IDS=zero: what TF's GPU gather does with such an id -- a row of zeros, no gradient.  Only where the
                # graph has such a row already (the Hpmn class masks id 0, code/hpmn.py:417-423): the ids are mapped onto it.
                # The reference's own Taobao files need this: their target rows carry btag == feature_size
                # (preprocess_taobao.py:48,131), one past the table.
                if os.environ.get("HPMN_OOB_IDS", "raise") == "zero" and not industry and lo >= 0:
                    ids = np.where(np.asarray(ids) >= feature_size, 0, ids)
                else:
                    raise ValueError("dataset ids span [%d, %d] but the embedding table has %d rows (tf.nn.embedding_lookup "
                                     "raises on the CPU; HPMN_OOB_IDS=zero gives its GPU behaviour for the Hpmn class: a "
                                     "zero row, no gradient)" % (lo, hi, feature_size))
        # int32 ids like the reference's placeholders (code/hpmn.py:248-251) -- int64 only for a table whose row count does not
        # fit them (BASELINE configs[4] sized to HBM; the documented deviation of SURVEY.md section 7, hard part 4)
        id_dtype = np.int64 if (feature_size is not None and feature_size > 2 ** 31 - 1) else np.int32
        self.ids = torch.as_tensor(np.ascontiguousarray(ids, dtype=id_dtype)).to(device)
        self.item_ids = None
        if want_item:
            if feature_size is not None and self.n and (int(np.min(item_ids)) < 0 or int(np.max(item_ids)) >= feature_size):
                if os.environ.get("HPMN_OOB_IDS", "raise") == "zero" and not industry and int(np.min(item_ids)) >= 0:
                    item_ids = np.where(np.asarray(item_ids) >= feature_size, 0, item_ids)
                else:
                    raise ValueError("item-side ids out of the embedding table's range")
            self.item_ids = torch.as_tensor(np.ascontiguousarray(item_ids, dtype=id_dtype)).to(device)
        self.label_np = np.asarray(label, dtype=np.int32)
        self.label = torch.as_tensor(self.label_np).to(device)
        self.length_np = None if length is None else np.asarray(length)

    def batches(self, batch_size: int):
        for lo in range(0, self.n, batch_size):
            yield lo, min(self.n, lo + batch_size)


class Hpmn_Basic(object):
    """Counterpart of ``Hpmn_Basic`` (code/hpmn.py:16-215): owns variables, optimiser state and
    the train step.  ``dist`` = (rank, world) enables data parallel over RCCL."""

    eval_every = 100
    industry = False
    _dp = False                  # data-parallel code paths on (set in __init__: world > 1, or forced collectives)
    compact_table_grad = False   # (set in __init__, _decide_compact_table_grad)
    _prefetched = None
    _flat_grad_clean = False
    _preset_plan = None
    _plan_stream = None

    def __init__(self, path, trainset, testset, feature_size, user_dim, item_dim, learning_rate,
                 hidden_size, embedding_size, hop, user_layers, item_layers, user_num_layers,
                 item_num_layers, user, item, emb_initializer=None, l2_reg=0, memory_reg=1e-5,
                 device=None, seed: Optional[int] = None, verbose: bool = True, lazy_table_adam: Optional[bool] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("hpmn_amd needs an MI355X (ROCm) device: the hot path is HIP-only, "
                               "there is no CPU fallback")
        ops._lib.load()       # raises loudly if libhpmn_hip.so is not built
        if not user and not item:
            raise ValueError("at least one of user / item must be set (code/hpmn.py:452-462)")
        self._path = path
        self.trainset, self.testset = trainset, testset
        self.feature_size = int(feature_size)
        self.learning_rate = float(learning_rate)
        self.l2_reg, self.memory_reg = float(l2_reg), float(memory_reg)
        self.hidden_size, self.embedding_size = int(hidden_size), int(embedding_size)
        self.hop = int(hop)
        self.user_layers, self.item_layers = list(user_layers), list(item_layers)
        self.user_num_layers, self.item_num_layers = int(user_num_layers), int(item_num_layers)
        self.user_dim, self.item_dim = int(user_dim), int(item_dim)
        self.user, self.item = user, item
        self.verbose = verbose
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        assert self.user_num_layers <= len(self.user_layers)     # code/hpmn.py:115
        assert not item or self.item_num_layers <= len(self.item_layers)
        self.rank, self.world = dist.rank_world()
        # data-parallel code paths on: more than one rank, or HPMN_DP_FORCE_COLLECTIVES=1 inside an initialised process
        # group of ONE rank (every collective really issued: how a 1-GPU box exercises the RCCL calls, tests/test_gpu_dp.py)
        self._dp = self.world > 1 or dist.forced()
        # Row-wise ("lazy") Adam on the embedding table: ONLY for tables that cannot afford the reference's dense
        # update (BASELINE configs[4]); a labelled deviation from code/hpmn.py:209-214, off unless asked for.
        self.lazy_table_adam = bool(int(os.environ.get("HPMN_LAZY_TABLE_ADAM", "0"))) if lazy_table_adam is None \
            else bool(lazy_table_adam)
        if self.lazy_table_adam and (item or l2_reg):
            raise NotImplementedError("lazy_table_adam: user-only graph, l2_reg == 0")
        self.table_exchange_chunks = int(os.environ.get("HPMN_TABLE_EXCHANGE_CHUNKS", "4"))
        # how the replicas keep the (replicated) table in step: "allreduce" = sum all-reduce of the table gradient in
        # a few ranges + replicated dense Adam (every rank sweeps the whole table); "sharded" = reduce-scatter of the
        # gradient, Adam on this rank's 1/world of the rows only, all-gather of the updated rows -- the same bytes
        # on the wire, 1/world of the 28 B/element optimiser traffic per GPU (identical arithmetic); "single" = one
        # blocking all-reduce over the whole flat gradient, then one update (no overlap: the fallback switch)
        # auto: the two-pass step with touched rows for big tables, dense all-reduce for small ones (_train_step_dp)
        self.table_exchange = os.environ.get("HPMN_TABLE_EXCHANGE", "auto")
        # Deterministic table gradients (r4): the scatter as a segmented reduction in row order, no atomics (ops.ScatterPlan,
        # csrc/scatter_sorted.hip).  HPMN_DET_SCATTER=0: the atomic kernel (run-length pre-reduced fp32 atomics).
        # "auto" (default): on under the data-parallel rows exchange, which needs the batch's distinct rows and their compact
        # gradient rows anyway (the plan replaces torch.unique + index_select there); off in a single process, where it costs
        # 3 % of the C3 step (2.81 -> 2.90 ms: the 240 us sort shares HBM with the forward, the two passes are 160 us beside
        # the weight gradient where the atomic kernel is 136) and 17 % of the 0.33 ms C1 step (~20 launches of host time).
        # "1": always (bit-reproducible training, tests/test_gpu_parity.py::test_training_steps_are_bit_reproducible).
        self._det_env = os.environ.get("HPMN_DET_SCATTER", "auto")
        self.det_scatter = self._det_env == "1"
        self._plan_wants_rows = False     # (the data-parallel rows exchange sends the plan's compact rows)
        self._plan_row_bounds = None
        self._plan_caps = (0, 0)          # (r5: capacities of the plan's `rows` / `out_rows` buffers as the exchange sends them)
        self._preset_plan = None          # (r5: the next compute_gradients takes this plan instead of building one)
        self._prefetched = None           # (r5: what train_step(next_ids=) prepared for the next step)
        self.last_scatter_plan = None
        self._sharded_moments = False     # set once the sharded table update has run (save_model gathers the moments then)
        self.TWO_PASS_MIN_NUMEL = int(os.environ.get("HPMN_TWO_PASS_MIN_NUMEL", str(type(self).TWO_PASS_MIN_NUMEL)))
        self.last_exchange_bytes = 0            # bytes this rank received in the last step's table exchange
        self._dropout_base, self._dropout_step = (int(seed or 0) * 0x632BE59BD9B4E019 + 0x1234567) & (2 ** 63 - 1), 0
        self._save_path = None
        self._datasets: Dict[int, Tuple[object, _DeviceDataset]] = {}
        self.spec = self._make_spec()
        self.item_spec = self._make_item_spec() if item else None
        # scope names of the two branches in the TF graph (code/hpmn.py:436, :444 "item"; :286, :297 "Item")
        self.item_scope = "Item" if self.industry else "item"
        self._branches = ([("User", self.spec, self.user_num_layers)] if user else []) + \
                         ([(self.item_scope, self.item_spec, self.item_num_layers)] if item else [])
        self._hip_read = bool(user and not item)     # user-only graph: the four-library-call step (no item-side scan)
        self.compact_table_grad = self._decide_compact_table_grad()
        self._build_variables(emb_initializer, seed)
        self.adam_t = 0
        self.beta1, self.beta2, self.adam_eps = 0.9, 0.999, 1e-8   # tf.train.AdamOptimizer defaults
        os.makedirs(self._path, exist_ok=True)

    def _decide_compact_table_grad(self) -> bool:
        """r5 (ABI v12): NO dense [V, E] gradient table.  The deterministic scatter leaves one summed gradient row per distinct
        table row of the batch (ops.ScatterPlan.out_rows), hpmn_rows_sum_adam updates exactly those rows -- all ranks' rows in
        one launch under data parallel -- and the early pass of the two-pass table Adam covers the rest: the arithmetic of
        code/hpmn.py:209-214's dense update with a quarter less table state (p, m, v: 3 x V x E floats instead of 4) and a
        third of the serial tail (C3: scatter 118 us + late pass 155 us -> segmented reduction + 56 us).
        HPMN_TABLE_GRAD: ``auto`` (default) = compact wherever the two-pass step applies (user-only graph, no l2 term, dense
        Adam, a table of >= TWO_PASS_MIN_NUMEL elements, and under data parallel the rows exchange on <= 8 ranks);
        ``dense`` = the r4 layout; ``compact`` = compact or raise."""
        mode = os.environ.get("HPMN_TABLE_GRAD", "auto")
        if mode not in ("auto", "dense", "compact"):
            raise ValueError("HPMN_TABLE_GRAD must be auto, dense or compact (got %r)" % mode)
        if self._det_env not in ("0", "1", "auto"):
            #  any other spelling used to switch the plan on in one place and off in another
            raise ValueError("HPMN_DET_SCATTER must be 0, 1 or auto (got %r)" % self._det_env)
        ok = bool(self._hip_read and not self.l2_reg and not self.lazy_table_adam and self.TWO_PASS_TABLE_ADAM
                  and self._table_adam_width_ok() and self._det_env != "0"
                  and (not self._dp or (self.table_exchange in ("auto", "rows") and self.world <= ops._lib.HPMN_MAX_RANKS)))
        if mode == "compact" and not ok:
            raise ValueError("HPMN_TABLE_GRAD=compact needs the user-only graph, l2_reg == 0, dense Adam, E/4 a power of two "
                             "<= 64, HPMN_DET_SCATTER != 0 and (data parallel) HPMN_TABLE_EXCHANGE auto / rows on <= 8 ranks")
        numel = self.feature_size * self.embedding_size
        # auto: wherever the plan is needed anyway (the data-parallel rows exchange) and where a fourth table-sized buffer is
        # what does not fit (tables from COMPACT_MIN_BYTES: configs[4]); in a single process on the reference tables the r4 step
        # stays -- measured r5 at C3 / C2: 2.56-2.62 vs 2.46-2.49 and 1.02-1.06 vs 0.95-0.97 ms/step, the tail there is bound by
        # layer 0's weight gradient and HBM, not by the scatter + late pass this form shortens, and the plan's sort costs its
        # 180 us of small launches somewhere under the scans
        big = numel >= self.TWO_PASS_MIN_NUMEL and (self._dp or 4 * numel >= self.COMPACT_MIN_BYTES)
        return ok and (mode == "compact" or (mode == "auto" and big))

    COMPACT_MIN_BYTES = int(os.environ.get("HPMN_COMPACT_MIN_BYTES", str(8 << 30)))

    # ------------------------------------------------------------------ graph description
    def _make_spec(self) -> ScanSpec:
        raise NotImplementedError

    def _make_item_spec(self) -> ScanSpec:
        raise NotImplementedError

    # ------------------------------------------------------------------ variables
    def _param_shapes(self) -> List[Tuple[str, Tuple[int, ...]]]:
        """The variables the graph EXECUTES (TF also creates the other branch's, which no fetch ever reaches)."""
        H = self.hidden_size
        shp = [("Embedding/emb_mtx", (self.feature_size, self.embedding_size))]
        width = 0
        # (every branch's GRU variables first, the read-path variables behind them: the read kernel addresses ITS
        #  variables -- of both branches in dual mode -- as one contiguous range of the flat buffer)
        for scope, spec, K in self._branches:
            D0 = spec.D0
            for i in range(K):
                d = D0 if i == 0 else H
                shp += [("%s/GRU%d/gates/kernel" % (scope, i), (d + H, 2 * H)), ("%s/GRU%d/gates/bias" % (scope, i), (2 * H,)),
                        ("%s/GRU%d/candidate/kernel" % (scope, i), (d + H, H)), ("%s/GRU%d/candidate/bias" % (scope, i), (H,))]
        for scope, spec, K in self._branches:
            D0 = spec.D0
            shp += [(scope + "/dense/kernel", (D0, H)), (scope + "/dense/bias", (H,)), (scope + "/map", (H, H))]
            n = 1
            for _ in range(self.hop):
                for fin, fout in ((4 * H, 80), (80, 40), (40, 1)):      # code/hpmn.py:137-139
                    shp += [("%s/dense_%d/kernel" % (scope, n), (fin, fout)), ("%s/dense_%d/bias" % (scope, n), (fout,))]
                    n += 1
            width += H + D0
        shp += [("output/bn1/gamma", (width,)), ("output/bn1/beta", (width,))]
        for name, fin, fout in (("fc1", width, 200), ("fc2", 200, 80), ("fc3", 80, 1)):   # :191-195
            shp += [("output/%s/kernel" % name, (fin, fout)), ("output/%s/bias" % name, (fout,))]
        return shp

    def _build_variables(self, emb_initializer, seed):
        """All variables live in ONE flat fp32 buffer (with matching flat grad / Adam m / v), so the
        optimiser is one kernel launch and data parallel is one all-reduce.  Initialisers are
        TF1.4's defaults: glorot-uniform kernels, zero biases, GRU gate bias 1.0."""
        shapes = self._param_shapes()
        offs, n = {}, 0
        for name, shape in shapes:
            offs[name] = n
            n += (int(np.prod(shape)) + 3) // 4 * 4      # keep every view 16-byte aligned
            if name == "Embedding/emb_mtx" and self.world > 1:
                q = 256 * self.world                     # the table region splits evenly over the ranks ("sharded")
                n = (n + q - 1) // q * q
        self._n_flat = n
        self._offs = offs
        dev = self.device
        self.flat_param = torch.zeros(n, device=dev, dtype=torch.float32)
        # lazy table Adam: no dense table gradient exists; the flat gradient then covers the dense variables only
        self._goff = offs[shapes[1][0]] if (self.lazy_table_adam or self.compact_table_grad) else 0
        self.flat_grad = torch.zeros(n - self._goff, device=dev, dtype=torch.float32)
        self._loss_acc = torch.zeros(2, device=dev, dtype=torch.float32)      # log-loss sum, memory-loss sum of a step
        self._loss_acc_clean = True
        # (r5) the plain train_step's Adam launches consume the gradient (hpmn_adam_step_clear): all-zero again behind them
        self._flat_grad_clean = True
        # two-pass dense table Adam (train_step): rows the batch points at / whether the table gradient is all-zero
        self._row_flags: Optional[torch.Tensor] = None
        self._table_grad_clean = True
        # housekeeping off the serial chain.  HPMN_AUX_PRIORITY=1 (r6, measurement switch): the auxiliary and plan streams at high
        # priority, i.e. in the hardware-queue pool of the library's helper streams instead of the caller's and RCCL's
        self._bg_priority = -1 if os.environ.get("HPMN_AUX_PRIORITY", "0") == "1" else 0
        self._aux_stream = torch.cuda.Stream(device=dev, priority=self._bg_priority)
        self.flat_m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(n, device=dev, dtype=torch.float32)
        self.params: Dict[str, torch.Tensor] = {}
        self.grads: Dict[str, torch.Tensor] = {}
        gen = torch.Generator(device=dev)
        gen.manual_seed(0 if seed is None else int(seed))
        for name, shape in shapes:
            k = int(np.prod(shape))
            p = self.flat_param[offs[name]:offs[name] + k].view(shape)
            g = None if ((self.lazy_table_adam or self.compact_table_grad) and name == "Embedding/emb_mtx") else \
                self.flat_grad[offs[name] - self._goff:offs[name] - self._goff + k].view(shape)
            if name == "Embedding/emb_mtx" and emb_initializer is not None:
                p.copy_(torch.as_tensor(np.asarray(emb_initializer, dtype=np.float32)))
            elif name.endswith("gates/bias") or name.endswith("gamma"):
                p.fill_(1.0)
            elif len(shape) == 2:
                _glorot_uniform_(p, gen)
            self.params[name] = p
            self.grads[name] = g
        self._emb_numel_padded = offs[shapes[1][0]]     # emb is first; dense part starts here
        self._gru_names_of = {scope: [["%s/GRU%d/%s" % (scope, i, s) for s in
                                       ("gates/kernel", "gates/bias", "candidate/kernel", "candidate/bias")]
                                      for i in range(K)] for scope, _, K in self._branches}
        self._gru_names = self._gru_names_of[self._branches[0][0]]
        self._make_read_desc()

    def set_params(self, values: Dict[str, np.ndarray]):
        """Inject weights (parity tests: identical weights => identical logits)."""
        with torch.no_grad():
            for k, v in values.items():
                self.params[k].copy_(torch.as_tensor(np.asarray(v, dtype=np.float32)).to(self.device))

    def get_params(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().cpu().numpy().copy() for k, v in self.params.items()}

    def _gru_weights(self, scope: Optional[str] = None) -> List[torch.Tensor]:
        names = self._gru_names if scope is None else self._gru_names_of[scope]
        return [self.params[n] for ns in names for n in ns]

    # ------------------------------------------------------------------ read path (code/hpmn.py:133-207)
    def _make_read_desc(self):
        """Offsets of the read-path variables inside their contiguous range of the flat buffer: one descriptor per branch
        the graph executes (user, item -- the order of the head's concat); the head's offsets ride in the first."""
        from ._lib import HpmnReadDesc
        first = self._branches[0][0]
        start = self._offs[first + "/dense/kernel"]
        rel = lambda name: self._offs[name] - start
        descs = []
        for scope, spec, K in self._branches:
            d = HpmnReadDesc()
            d.K, d.H, d.D0, d.hop = K, self.hidden_size, spec.D0, self.hop
            d.off_wq, d.off_bq, d.off_map = rel(scope + "/dense/kernel"), rel(scope + "/dense/bias"), rel(scope + "/map")
            for h in range(self.hop):
                for j in range(3):
                    d.off_att[h][2 * j] = rel("%s/dense_%d/kernel" % (scope, 3 * h + j + 1))
                    d.off_att[h][2 * j + 1] = rel("%s/dense_%d/bias" % (scope, 3 * h + j + 1))
            d.off_gamma, d.off_beta = rel("output/bn1/gamma"), rel("output/bn1/beta")
            for j, name in enumerate(("fc1", "fc2", "fc3")):
                d.off_fc[2 * j] = rel("output/%s/kernel" % name)
                d.off_fc[2 * j + 1] = rel("output/%s/bias" % name)
            d.n_params = self._n_flat - start
            descs.append(d)
        self._read_descs = descs
        self._read_desc = descs[0]
        self._read_params = self.flat_param[start:]
        self._read_grads = self.flat_grad[start - self._goff:]

    # ------------------------------------------------------------------ forward passes
    @torch.no_grad()
    def forward_inference(self, ids: torch.Tensor, want_logit=True, want_att=True, item_ids: Optional[torch.Tensor] = None):
        """Eval-mode forward (keep_prob 1): hpmn_scan_fwd chain + hpmn_read_fwd."""
        if not self._hip_read:
            return self._forward_inference_branches(ids, item_ids)
        if self._tiled_inference(ids.shape[0]):
            # evaluation-sized batches: the 16-sequence-tile MFMA kernel, layer by layer (ops.tiled_forward_inference)
            memory, last = ops.tiled_forward_inference(self.spec, ids, self.params["Embedding/emb_mtx"], self._gru_weights(),
                                                       group=1)
        else:
            memory, last = ops.scan_forward_inference(self.spec, ids, self.params["Embedding/emb_mtx"],
                                                      self._gru_weights())
        if ids.shape[0] == 0:
            z = torch.empty(0, device=self.device)
            return dict(memory=memory, prediction=z, logit=z, user_weights=torch.empty(0, self.spec.K, device=self.device),
                        memory_loss=torch.zeros((), device=self.device))
        return ops.read_fwd(self._read_desc, self._read_params, memory, last, want_logit, want_att)

    # Rows from which the forward-only path switches from one sequence per wave to 16-sequence tiles on the matrix cores
    # (r4, C3 shape, build_memory alone: 673 k / 565 k sequences/s at 1000 rows, 690 k / 1.06 M at 2000, 725 k / 1.76 M at
    # 4000 -- a tile kernel needs ~256 tiles to fill the chip, the per-sequence kernels saturate at one wave per SIMD)
    TILED_EVAL_MIN_ROWS = int(os.environ.get("HPMN_TILED_EVAL_MIN_ROWS", "1536"))
    TILED_EVAL_ROWS = int(os.environ.get("HPMN_TILED_EVAL_ROWS", "4096"))        # rows eval() puts in flight per pass

    def _tiled_inference(self, rows: int) -> bool:
        return bool(self.TILED_EVAL_MIN_ROWS > 0 and rows >= self.TILED_EVAL_MIN_ROWS and self.spec.H in (64, 128)
                    and self.spec.E % 4 == 0 and ops.tile_kernel_supported(self.spec.H, self.spec.D0))

    # ------------------------------------------------------------------ graphs that execute the item branch
    def _branch_inputs(self, ids, item_ids):
        out = []
        for scope, spec, _ in self._branches:
            x = ids if scope == "User" else item_ids
            if x is None:
                raise ValueError("this graph executes the %s branch: pass its id tensor" % scope)
            out.append((scope, spec, x))
        return out

    @torch.no_grad()
    def _forward_inference_branches(self, ids, item_ids):
        """item=True graphs, eval mode: both scans on the HIP inference chain, the joint read path -- two attention stacks
        into one head -- in ONE launch of the read kernel (hpmn_read_fwd_n)."""
        emb = self.params["Embedding/emb_mtx"]
        scopes, memories, lasts = [], [], []
        for scope, spec, x in self._branch_inputs(ids, item_ids):
            memory, last = ops.scan_forward_inference(spec, x, emb, self._gru_weights(scope))
            scopes.append(scope); memories.append(memory); lasts.append(last)
        mems = dict(zip(scopes, memories))
        B = memories[0].shape[0]
        if B == 0:
            z = torch.empty(0, device=self.device)
            return dict(prediction=z, logit=z, memory_loss=torch.zeros((), device=self.device), memory=memories[0],
                        user_weights=torch.empty(0, self._branches[0][2], device=self.device))
        out = ops.read_fwd_n(self._read_descs, self._read_params, memories, lasts)
        w = dict(zip(scopes, out["weights"]))
        return dict(prediction=out["prediction"], logit=out["logit"], memory_loss=out["memory_loss"],
                    memory=memories[0], memories=mems, user_weights=w.get("User"), item_weights=w.get(self.item_scope))

    def _compute_gradients_branches(self, ids, item_ids, label, keep_prob, masks, global_batch, defer_join):
        """item=True graphs: scan forward (per branch) -> the joint read path, its loss and every gradient of them in one
        launch of the read kernel (hpmn_read_fwd_bwd_n) -> scan BPTT + embedding scatter (per branch) into the flat
        gradient.  No autograd anywhere."""
        emb = self.params["Embedding/emb_mtx"]
        fw = []
        with torch.no_grad():
            for scope, spec, x in self._branch_inputs(ids, item_ids):
                memory, last, saved = ops.scan_forward_train(spec, x, emb, self._gru_weights(scope))
                fw.append((scope, spec, x, memory, last, saved))
            seed = 0
            if masks is None and keep_prob < 1.0:
                self._dropout_step += 1
                seed = _splitmix64(_splitmix64(self._dropout_base + self._dropout_step) ^ (self.rank + 1)) | 1
            out = ops.read_fwd_bwd_n(self._read_descs, self._read_params, self._read_grads, [f[3] for f in fw],
                                     [f[4] for f in fw], label, masks, keep_prob, 1.0 / float(global_batch),
                                     self.memory_reg, dropout_seed=seed)
            ll_sum = out["log_loss_sum"]
            for j, (scope, spec, x, memory, last, saved) in enumerate(fw):
                grad_out = [self.grads["Embedding/emb_mtx"]] + [self.grads[n] for ns in self._gru_names_of[scope] for n in ns]
                pend = ops.scan_backward(spec, x, saved, self._gru_weights(scope), out["d_memory"][j], out["d_last"][j],
                                         grad_out, defer_join=False)
                if pend is not None:
                    pend.join()
            res = dict(prediction=out["prediction"].detach(), log_loss_sum=ll_sum.detach(),
                       memory_loss=out["memory_loss"].detach(), memory=fw[0][3], pending=None)
            if self.l2_reg:
                self.flat_grad.add_(self.flat_param, alpha=self.l2_reg / self.world)
            ce = res["log_loss_sum"] / float(global_batch) + self.memory_reg * res["memory_loss"]
            if self.l2_reg:
                ce = ce + (0.5 * self.l2_reg / self.world) * sum((v * v).sum() for v in self.params.values())
        return res, ce

    @torch.no_grad()
    def compute_gradients(self, ids: torch.Tensor, label: torch.Tensor, keep_prob=0.5, masks=None,
                          global_batch: Optional[int] = None, defer_join: bool = False,
                          item_ids: Optional[torch.Tensor] = None, _clear_grads=None):
        """Forward + BPTT of cross_entropy (code/hpmn.py:202-207) for a (possibly sharded) batch into
        the flat gradient buffer: log-loss is a MEAN over the GLOBAL batch, the memory regulariser a
        SUM.  Pure kernel sequence: scan fwd -> read fwd+loss+bwd -> scan bwd.
        With ``defer_join`` the GRU weight gradients may still be running on a side stream:
        out["pending"].join() must be called before they (or anything after the table in the flat
        gradient) are read; the embedding-table gradient is complete on the current stream."""
        B = ids.shape[0]
        if global_batch is None:
            global_batch = B * self.world
        grad_was_clean, self._flat_grad_clean = self._flat_grad_clean, False
        if B == 0 or not self._hip_read:
            self.flat_grad.zero_()
            self._table_grad_clean = True
            # (an empty shard leaves the flag False: under data parallel the all-reduces that follow write the OTHER ranks'
            #  gradient sum into this buffer, and only the branch of train_step whose Adam launches clear every range sets
            #  the flag again -- ADVICE r5)
        if B == 0:
            return dict(prediction=torch.empty(0, device=self.device)), torch.zeros((), device=self.device)
        if not self._hip_read:
            return self._compute_gradients_branches(ids, item_ids, label, keep_prob, masks, global_batch, defer_join)
        # Housekeeping that nothing on the serial chain waits for goes to an auxiliary stream: clearing the flat
        # gradient (213 MB at C3, 28 us) runs underneath the forward scans, which do not touch it, and the three
        # scalar kernels that form cross_entropy run underneath BPTT instead of in front of the table update.
        # (Each hand-over between streams costs a few microseconds of queue processing: only worth it where the
        # gradient buffer is big -- C3: 213 MB -- not for the 0.5 ms steps of the small-table configurations.)
        main = torch.cuda.current_stream()
        aux = self._aux_stream if (self.flat_grad.numel() >= self.AUX_MIN_NUMEL or self.compact_table_grad) else main
        if aux is not main:
            aux.wait_stream(main)                            # (after the previous step's optimiser, which read it)
        cleared = None
        rest2 = None
        # The deterministic scatter's row order (ops.ScatterPlan: a stable sort of the batch's ids) depends on the ids alone:
        # built on the auxiliary stream, consumed behind BPTT.  WHERE on that stream matters (r4: in front of the early
        # table-Adam pass its 240 us delayed that pass past the forward and cost the C3 step 0.26 ms): behind the pass, unless
        # the data-parallel rows exchange wants the distinct-row count at the start of the step.
        plan, plan_ready = None, None
        det = self.det_scatter or (self._det_env == "auto" and self._plan_wants_rows) or self.compact_table_grad
        det = bool(det and not self.lazy_table_adam and self._table_adam_width_ok())     # (E/4 a power of two <= 64: ADVICE r4)
        want_rows = bool(self._plan_wants_rows or self.compact_table_grad)

        def make_plan():
            if self._preset_plan is not None:                # (r5: prepared underneath the previous step, train_step(next_ids=))
                pl, self._preset_plan = self._preset_plan, None
                self.last_scatter_plan = pl
                return pl
            if os.environ.get("HPMN_PLAN_CACHE_EXPERIMENT") == "1":      # (measurement only: what a free plan would give)
                key = (ids.data_ptr(), tuple(ids.shape), self._plan_caps, tuple(self._plan_row_bounds or ()))
                hit = self.__dict__.setdefault("_plan_cache", {}).get(key)
                if hit is not None:
                    hit.ready = torch.cuda.Event()
                    hit.ready.record(self._aux_stream)
                    self.last_scatter_plan = hit
                    return hit
            pl = make_plan_()
            if os.environ.get("HPMN_PLAN_CACHE_EXPERIMENT") == "1":
                self._plan_cache[key] = pl
            return pl

        def make_plan_():
            pst = self._aux_stream
            if pst != torch.cuda.current_stream():
                pst.wait_stream(main)                        # (the ids may have been produced on the caller's stream just now)
            with torch.cuda.stream(pst):
                pl = ops.ScatterPlan(ids, self.embedding_size, want_rows=want_rows, host_count=False,
                                     row_bounds=self._plan_row_bounds if self._plan_wants_rows else None,
                                     rows_capacity=self._plan_caps[0], out_rows_capacity=self._plan_caps[1],
                                     V=self.feature_size)
                pl.ready = torch.cuda.Event()
                pl.ready.record(pst)
            pl.record_stream(main)
            self.last_scatter_plan = pl
            return pl
        if det and self._plan_wants_rows:
            plan = make_plan()
        with torch.cuda.stream(aux):
            rest = None
            if _clear_grads is not None:
                rest = _clear_grads()                        # train_step's two-pass table update (see there)
            elif not grad_was_clean:
                self.flat_grad.zero_()                       # (the plain train_step leaves it all-zero: no launch here then)
            if not self._loss_acc_clean:
                self._loss_acc.zero_()                       # (normally the previous step's reduce launch has cleared it)
            if callable(rest):
                # what the read kernel waits for is the CLEARING; the early table-Adam pass behind it is only needed in
                # front of the late pass (the wait at the end of this function) -- at C2 it outlasts the forward by 150 us
                cleared = torch.cuda.Event()
                cleared.record(aux)
                rest2 = None
                if not (self.EARLY_PASS_BESIDE_L0_REVERSE and aux is not main):
                    rest2 = rest()
                    rest = None
            if det and plan is None:
                plan = make_plan()
        if plan is not None:
            plan_ready = plan.ready
        self._table_grad_clean = False                       # (until something consumes or clears the table gradient)
        emb = self.params["Embedding/emb_mtx"]
        weights = self._gru_weights()
        probe = self._split_probe if isinstance(self._split_probe, dict) and self._split_probe.get("armed") else None
        if probe is not None:
            probe["f0"].record()
        memory, last, saved = ops.scan_forward_train(self.spec, ids, emb, weights)
        if probe is not None:
            probe["f1"].record()
        if aux is not main:
            if cleared is not None:
                main.wait_event(cleared)
            else:
                main.wait_stream(aux)
        seed = 0
        if masks is None and keep_prob < 1.0:
            # masks are drawn inside the read kernel (counter-based): a fresh 64-bit seed per step and rank
            self._dropout_step += 1
            seed = _splitmix64(_splitmix64(self._dropout_base + self._dropout_step) ^ (self.rank + 1)) | 1
        out = ops.read_fwd_bwd(self._read_desc, self._read_params, self._read_grads, memory, last, label, masks,
                               keep_prob, 1.0 / float(global_batch), self.memory_reg, dropout_seed=seed,
                               loss_out=self._loss_acc, defer_param_grads=True)
        self._loss_acc_clean = False
        sums = (torch.empty if ids.shape[0] > 0 else torch.zeros)(3, device=self.device, dtype=torch.float32)

        def read_param_grads():
            # the read path's weight gradients (only the optimiser needs them: off the serial chain where there is an
            # auxiliary stream) and, from the same two launches, the loss scalars + the cleared accumulator
            with torch.cuda.stream(aux):
                out.pop("reduce_param_grads")(sums)
                self._loss_acc_clean = True
                if aux is not main:
                    sums.record_stream(main)
                if callable(rest2):
                    rest2()                                  # (second part of the early table-Adam pass: beside BPTT)

        # (small tables keep their housekeeping on the caller's stream: there the two launches go BEHIND the reverse scans
        #  -- r5: in front of them they were 15 us of the Amazon step's serial chain, behind them they fill the slack the
        #  table update's chain has against the GRU weight gradients' on the helper stream)
        if aux is not main:
            aux.wait_stream(main)
            read_param_grads()
        out["log_loss_sum"], out["memory_loss"], ce = sums[0], sums[1], sums[2]
        if self.lazy_table_adam:
            # touched rows only: the scatter goes to a COMPACT [U, E] buffer through ids remapped to 0..U-1 (row 0 of
            # it stays original id 0, so the id-0 mask of the Hpmn graph keeps working on the remapped ids)
            flat = ids.reshape(-1).long()
            if self.spec.mask_id0:
                flat = torch.cat([flat.new_zeros(1), flat])
            uniq, inv = torch.unique(flat, return_inverse=True)
            if self.spec.mask_id0:
                inv = inv[1:]
            d_emb = torch.zeros(uniq.numel(), self.embedding_size, device=self.device, dtype=torch.float32)
            scatter_ids = inv.to(torch.int32).view_as(ids).contiguous()
            out["table_rows"], out["table_row_grads"] = uniq, d_emb
        else:
            d_emb, scatter_ids = self.grads["Embedding/emb_mtx"], ids
        grad_out = [d_emb] + [self.grads[n] for names in self._gru_names for n in names]
        if callable(rest):
            ops.train_mark_layer0_reverse(self.device, True)
        if plan_ready is not None:
            main.wait_event(plan_ready)
        pending = ops.scan_backward(self.spec, scatter_ids, saved, weights, out["d_memory"], out["d_last"], grad_out,
                                    defer_join=defer_join and not self.l2_reg, scatter_plan=plan)
        out["pending"] = pending
        if aux is main:
            read_param_grads()
        if callable(rest):
            # HPMN_EARLY_PASS=bwd: the early table-Adam pass beside layer 0's REVERSE launch instead of beside its forward
            # (which can then be one of the two-layer launches, HPMN_PAIR_FWD=1)
            ops.train_wait_layer0_reverse(self.device, aux)
            with torch.cuda.stream(aux):
                rest()
        if aux is not main:
            main.wait_stream(aux)                            # the loss scalars belong to the caller's stream again
        if self.l2_reg:
            # l2_reg * tf.nn.l2_loss(v) for every trainable variable (code/hpmn.py:204-205); every rank holds
            # every variable, so each adds 1/world of it before the sum all-reduce
            self.flat_grad.add_(self.flat_param, alpha=self.l2_reg / self.world)
        if self.l2_reg:
            # cross_entropy includes sum_v l2_reg * tf.nn.l2_loss(v) = l2_reg/2 * |v|^2 (code/hpmn.py:203-205);
            # every rank holds every variable, so each reports 1/world of it like the other (sharded) terms
            ce = ce + (0.5 * self.l2_reg / self.world) * sum((v * v).sum() for v in self.params.values())
        out["memory"] = memory
        return out, ce

    # ------------------------------------------------------------------ one training step
    def train_step(self, ids: torch.Tensor, label: torch.Tensor, keep_prob=0.5, masks=None,
                   global_batch: Optional[int] = None, item_ids: Optional[torch.Tensor] = None,
                   next_ids: Optional[torch.Tensor] = None, next_global_batch: Optional[int] = None):
        """sess.run(train_step) of code/hpmn.py:482: forward, BPTT, clip, dense TF Adam.
        ``next_ids`` (optional, a hint): the ids tensor the NEXT call will be given (and, data parallel, its global batch) --
        what only depends on the ids (the scatter's plan; under data parallel the exchange of the ranks' distinct rows and
        counts) is then prepared underneath this step's BPTT instead of in front of the next step's early table pass.  Every
        rank must pass it or none (it issues collectives)."""
        if self.auto_det_distinct_fraction is None and ids.shape[0] > 0:
            self._probe_id_law(ids, set_hint=True)
        if self.compact_table_grad:
            return self._train_step_rows(ids, label, keep_prob, masks, global_batch, next_ids, next_global_batch)
        if item_ids is None and self._one_call_ok(ids):
            return self._train_step_one_call(ids, label, keep_prob, masks, global_batch)
        if self._two_pass_table_adam(ids):
            return self._train_step_two_pass(ids, label, keep_prob, masks, global_batch)
        if self._dp_two_pass(ids):
            return self._train_step_dp(ids, label, keep_prob, masks, global_batch)
        out, ce = self.compute_gradients(ids, label, keep_prob, masks, global_batch, defer_join=True, item_ids=item_ids)
        pending = out.pop("pending", None)
        if self.l2_reg:
            # (same on every rank) the l2 term touched the whole buffer after the join: one exchange, one update
            dist.allreduce_sum_(self.flat_grad)                 # RCCL sum; clip happens after (8e)
            self.apply_gradients()
            return out, ce
        # From here on EVERY rank issues the same collectives in the same order, whatever its shard looked like
        # (a rank whose slice of a short last batch is empty has pending == None and a zero gradient).
        # The table gradient is final once the scatter is enqueued, the GRU weight gradients of layer 0
        # are still being reduced on the side stream: exchange + update the table (99.5 % of the
        # parameters, HBM-bound) underneath them, then join and do the dense rest.
        n_emb = self.params["Embedding/emb_mtx"].numel()
        if self.lazy_table_adam:
            self.adam_t += 1
            t = self.adam_t
            lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
            V, E = self.feature_size, self.embedding_size
            rows = out.pop("table_rows", None)
            row_grads = out.pop("table_row_grads", None)
            if self._dp:
                # every rank's (touched rows, their gradient rows) -> the union and the summed rows, identically on
                # every rank; an empty shard contributes nothing but takes part in the collectives
                if rows is None:
                    rows = torch.empty(0, device=self.device, dtype=torch.int64)
                    row_grads = torch.empty(0, E, device=self.device, dtype=torch.float32)
                counts = dist.exchange_counts(rows.numel(), self.device)
                wide = self.feature_size > 2 ** 31 - 1
                ids_all, g_all = dist.exchange_rows(rows, row_grads, counts, wide_ids=wide)
                self.last_exchange_bytes = dist.rows_exchange_bytes(counts, E, wide)
                valid = torch.cat([ids_all[r, :n] for r, n in enumerate(counts)]).long()
                rows = torch.unique(valid)
                row_grads = torch.zeros(rows.numel(), E, device=self.device, dtype=torch.float32)
                dist.sum_rows_into_(row_grads, ids_all, g_all, counts, row_of=lambda i: torch.searchsorted(rows, i))
            if rows is not None and rows.numel() > 0:
                ops.adam_step_rows(self.flat_param[:n_emb].view(V, E), row_grads,
                                   self.flat_m[:n_emb].view(V, E), self.flat_v[:n_emb].view(V, E), rows,
                                   lr_t, self.beta1, self.beta2, self.adam_eps, clip=1.0)
            if pending is not None:
                pending.join()
            lo = self._goff
            dist.allreduce_sum_(self.flat_grad)                 # (lazy: the flat gradient holds the dense variables only)
            ops.adam_step(self.flat_param[lo:], self.flat_grad, self.flat_m[lo:], self.flat_v[lo:], lr_t, self.beta1,
                          self.beta2, self.adam_eps, clip=1.0)
            return out, ce
        if self._dp and self.table_exchange == "single":
            # the plainest scheme (fallback switch): join, ONE all-reduce over the whole flat gradient, one update
            if pending is not None:
                pending.join()
            dist.allreduce_sum_(self.flat_grad)
            self.apply_gradients()
            return out, ce
        if self._dp and self.table_exchange == "sharded":
            n_pad = self._emb_numel_padded
            shard = n_pad // self.world
            lo = self.rank * shard
            self._sharded_moments = True
            g = dist.reduce_scatter_sum(self.flat_grad[:n_pad], self.rank, self.world)      # [shard]
            self.adam_t += 1
            t = self.adam_t
            lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
            ops.adam_step(self.flat_param[lo:lo + shard], g, self.flat_m[lo:lo + shard], self.flat_v[lo:lo + shard],
                          lr_t, self.beta1, self.beta2, self.adam_eps, clip=1.0)
            dist.all_gather_shards_(self.flat_param[:n_pad], lo, shard)
            n_emb = n_pad
        elif self._dp:
            # the table exchange is the one big collective of the step (212 MB at C3): cut it into a few
            # ranges so that clip + Adam of range i run while RCCL is still reducing range i+1
            bounds = dist.chunk_bounds(n_emb, self.table_exchange_chunks, align=1024)
            works = [dist.allreduce_sum_async(self.flat_grad[a:b]) for a, b in bounds]
            for i, ((a, b), w) in enumerate(zip(bounds, works)):
                w.wait()                                     # orders the current stream after the collective
                self.apply_gradients(a, b, advance=(i == 0), clear=True)
        else:
            self.apply_gradients(0, n_emb, advance=True, clear=True)
        if pending is not None:
            pending.join()
        dist.allreduce_sum_(self.flat_grad[n_emb:])
        self.apply_gradients(n_emb, self.flat_param.numel(), advance=False, clear=True)
        # (every element of the flat gradient has been consumed by a clearing Adam launch -- unless the table went the sharded way)
        self._flat_grad_clean = not (self._dp and self.table_exchange == "sharded")
        return out, ce

    # ------------------------------------------------------------------ the whole step behind ONE library call (r6)
    ONE_CALL_STEP = os.environ.get("HPMN_ONE_CALL_STEP", "1") != "0"
    _one_call_cache = None
    _phase_probe = None          # a list: the data-parallel rows step appends (start, BPTT enqueued, table updated, done) timing events

    def _one_call_ok(self, ids) -> bool:
        """The plain single-process step (one dense sweep over the table, everything on the caller's stream + the library's
        helper stream) is one call of hpmn_train_step: the graphs and settings for which train_step would otherwise issue
        exactly hpmn_scan_fwd_train -> hpmn_read_fwd_bwd -> hpmn_scan_bwd -> read weight gradients -> two clearing Adam launches.
        (The static part of the answer is cached per batch shape: the check itself was 8 us of a 60 us step.)"""
        if not self.ONE_CALL_STEP or self._preset_plan is not None or (self._split_probe.__class__ is dict and self._split_probe.get("armed")):
            return False
        key = (ids.shape, ids.dtype)
        ok = self._one_call_static.get(key)
        if ok is None:
            ok = bool(self._hip_read and not self._dp and not self.lazy_table_adam and not self.l2_reg and self._goff == 0
                      and ids.shape[0] > 0 and len(self._branches) == 1 and not self.det_scatter
                      and self.flat_grad.numel() < self.AUX_MIN_NUMEL and not self._two_pass_table_adam(ids)
                      and ops.TRAIN_ABI and ops.PIPELINE_CHUNKS <= 1 and ops.FUSED_FWD and not ops.SPLIT_LAYER0_BWD
                      and ops.pipe_mode(self.spec) == "")
            if self.__dict__.get("_one_call_static") is None:
                self._one_call_static = {}
            self._one_call_static[key] = ok
        return ok

    _one_call_static = {}

    @torch.no_grad()
    def _train_step_one_call(self, ids, label, keep_prob, masks, global_batch):
        """sess.run(train_step) of code/hpmn.py:482 as ONE library call (hpmn_train_step, ABI v14): the descriptor is built once
        per batch shape; a step fills in the batch's pointers and scalars.  Host work per step: two small allocations, a
        dozen struct stores, one ctypes call (tools/host_enqueue_time.py)."""
        B = ids.shape[0]
        if global_batch is None:
            global_batch = B
        stream = torch.cuda.current_stream().cuda_stream
        key = (B, ids.dtype, stream, self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.flat_m.data_ptr())
        cache = self._one_call_cache
        if cache is None or cache["key"] != key:
            from . import _lib
            import ctypes as C
            ops._chk_ids(ids)
            spec, dev = self.spec, self.device
            st = _lib.HpmnTrainStep()
            st.scan = spec.desc(B, self.feature_size, ids)
            C.memmove(C.byref(st.read), C.byref(self._read_desc), C.sizeof(_lib.HpmnReadDesc))
            st.read.B = B
            st.param, st.grad = self.flat_param.data_ptr(), self.flat_grad.data_ptr()
            st.m, st.v = self.flat_m.data_ptr(), self.flat_v.data_ptr()
            st.n_emb = self.params["Embedding/emb_mtx"].numel()
            st.n_total = self.flat_param.numel()
            for i, names in enumerate(self._gru_names):
                for j, n in enumerate(names):
                    st.off_gru[i][j] = self._offs[n]
            st.off_read = self._offs[self._branches[0][0] + "/dense/kernel"]
            lib = _lib.load()
            ws = torch.empty(int(lib.hpmn_scan_train_workspace_bytes(C.byref(st.scan))), device=dev, dtype=torch.uint8)
            self._read_desc.B = B
            keep = dict(ws=ws, memory=torch.empty(B, spec.K, spec.H, device=dev), last=torch.empty(B, spec.D0, device=dev),
                        d_memory=torch.empty(B, spec.K, spec.H, device=dev), d_last=torch.empty(B, spec.D0, device=dev),
                        rws=ops._read_workspace(self._read_desc, dev), acc=self._loss_acc)   # (zero-initialised once, then reused)
            st.memory, st.last = keep["memory"].data_ptr(), keep["last"].data_ptr()
            st.d_memory, st.d_last = keep["d_memory"].data_ptr(), keep["d_last"].data_ptr()
            st.scan_workspace, st.read_workspace = ws.data_ptr(), keep["rws"].data_ptr()
            st.loss_acc = self._loss_acc.data_ptr()
            st.memory_reg = float(self.memory_reg)
            st.beta1, st.beta2, st.eps, st.clip = self.beta1, self.beta2, self.adam_eps, 1.0
            st.scan.mask_id0 = self.spec.id_flags(ids)
            cache = self._one_call_cache = dict(key=key, st=st, keep=keep, ctx=ops._ctx(dev), fn=lib.hpmn_train_step, ref=C.byref(st),
                                                check=_lib.check)
        st = cache["st"]
        if label.dtype != torch.int32 or label.shape[0] != B or not label.is_contiguous() or not ids.is_contiguous():
            raise ValueError("train_step: ids [B,T,F] and label [B] int32, contiguous")
        seed = 0
        if masks is None:
            if keep_prob < 1.0:
                self._dropout_step += 1
                seed = _splitmix64(_splitmix64(self._dropout_base + self._dropout_step) ^ (self.rank + 1)) | 1
            st.mask1 = st.mask2 = None
        else:
            m1, m2 = masks
            ops._chk_f32(m1, m2)
            st.mask1, st.mask2 = m1.data_ptr(), m2.data_ptr()
        st.read.dropout_seed = seed & 0xFFFFFFFFFFFFFFFF
        st.ids, st.label = ids.data_ptr(), label.data_ptr()
        pred = torch.empty(B, device=self.device)
        sums = torch.empty(3, device=self.device)
        st.pred, st.loss3 = pred.data_ptr(), sums.data_ptr()
        st.keep_prob, st.inv_global_batch = keep_prob, 1.0 / global_batch
        if not self._loss_acc_clean:
            self._loss_acc.zero_()
        st.clear_grad_first = 0 if self._flat_grad_clean else 1
        self.adam_t = t = self.adam_t + 1
        st.lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        self._flat_grad_clean = self._table_grad_clean = self._loss_acc_clean = False
        rc = cache["fn"](cache["ctx"], cache["ref"], stream)
        if rc != 0:
            cache["check"](rc, "hpmn_train_step")
        self._flat_grad_clean = self._table_grad_clean = self._loss_acc_clean = True
        # (memory: the step's persistent scratch -- valid until the next train_step of this shape)
        ll, ml, ce = sums.unbind(0)
        return dict(prediction=pred, log_loss_sum=ll, memory_loss=ml, memory=cache["keep"]["memory"], pending=None), ce

    # ------------------------------------------------------------------ dense table Adam in two passes
    TWO_PASS_TABLE_ADAM = int(os.environ.get("HPMN_TWO_PASS_ADAM", "1")) != 0
    TWO_PASS_MIN_NUMEL = 1 << 24          # below this the dense sweep is a few microseconds: not worth two launches
    EARLY_PASS_BESIDE_L0_REVERSE = os.environ.get("HPMN_EARLY_PASS", "fwd") == "bwd"
    # fraction of the table rows whose early pass runs beside the forward (the rest: beside BPTT); "auto": measured once, on the
    # fourth step, from the durations of the forward and of the pass (HIP events) -- 1.0 unless the pass outlasts the forward
    EARLY_PASS_SPLIT = 1.0 if os.environ.get("HPMN_EARLY_SPLIT", "auto") == "auto" else float(os.environ["HPMN_EARLY_SPLIT"])
    EARLY_PASS_SPLIT_AUTO = os.environ.get("HPMN_EARLY_SPLIT", "auto") == "auto"
    AUX_MIN_NUMEL = int(os.environ.get("HPMN_AUX_MIN_NUMEL", str(1 << 24)))   # smaller gradient buffers: housekeeping stays on the caller's stream

    _split_probe = None
    _split_steps = 0

    def _tune_early_split(self):
        """EARLY_PASS_SPLIT = "auto": time the forward and the early pass of the fourth two-pass step with events, read them
        two steps later (one host wait, once) and keep what fits beside the forward."""
        if not self.EARLY_PASS_SPLIT_AUTO or self._split_probe == "done":
            return
        self._split_steps += 1
        if self._split_steps == 4:
            ev = lambda: torch.cuda.Event(enable_timing=True)
            self._split_probe = dict(armed=True, f0=ev(), f1=ev(), a0=ev(), a1=ev())
        elif self._split_steps == 5 and isinstance(self._split_probe, dict):
            self._split_probe["armed"] = False
        elif self._split_steps == 6 and isinstance(self._split_probe, dict):
            p = self._split_probe
            p["f1"].synchronize()
            p["a1"].synchronize()
            t_fwd, t_pass = p["f0"].elapsed_time(p["f1"]), p["a0"].elapsed_time(p["a1"])
            if t_pass > 1.05 * t_fwd > 0:
                self.EARLY_PASS_SPLIT = max(0.25, min(1.0, 0.85 * t_fwd / t_pass))
            self._split_probe = "done"

    def _table_adam_width_ok(self) -> bool:
        """hpmn_adam_step_table handles rows of E floats with E/4 a power of two <= 64 (api.hip); other widths take the
        dense one-sweep Adam."""
        e4, rem = divmod(self.embedding_size, 4)
        return rem == 0 and 1 <= e4 <= 64 and (e4 & (e4 - 1)) == 0

    # r6: the id law of a batch and the scatter.  The atomic kernel adds every RUN of equal ids along t with 16 float atomics;
    # under a heavy-tailed law -- Zipf(1.1) over the XLong item range puts 12 % of a batch's lookups on ONE row -- those
    # serialise in one L2 channel: C3 3.01 ms/step against 2.45 on uniform ids.  Measured alternatives on the same Zipf batches:
    # the sorted-segment reduction (every distinct row added once; its plan is ~0.2 ms of small launches) 2.51; the atomic
    # kernel with an LDS table per wave over equal ids (HPMN_ID_HOT) 2.40.  On uniform ids the table has nothing to gain and
    # makes the scatter launch 105 -> 133 us.  So the model looks at its FIRST training batch once (one torch.unique, one host
    # synchronisation in its life): distinct rows per run of equal ids below one half -> the hint is set.
    auto_det_distinct_fraction = None
    _auto_det_decision = None
    _hot_hint_mode = os.environ.get("HPMN_SCATTER_HOT_HINT", "auto")

    def _probe_id_law(self, ids, set_hint: bool = False) -> float:
        """Distinct rows per RUN of equal ids along t (what the atomic scatter issues is one row add per run -- the constant uid
        column of a sequence is one run): ~0.93 on uniform XLong ids, 0.20 on Zipf(1.1), 0.24 on the Taobao shape."""
        runs = int((ids[:, 1:] != ids[:, :-1]).sum()) + ids.shape[0] * ids.shape[2]
        self.auto_det_distinct_fraction = int(torch.unique(ids).numel()) / float(max(1, runs))
        if set_hint and self._hot_hint_mode == "auto":
            for _, spec, _ in self._branches[:1]:
                object.__setattr__(spec, "hot_ids", self.auto_det_distinct_fraction < 0.5)   # (frozen dataclass: a hint, not the graph)
            self._one_call_cache = None                       # (its descriptor carries the id flags)
        return self.auto_det_distinct_fraction

    def _two_pass_table_adam(self, ids) -> bool:
        """Single process, user-only graph, no densifying l2 term, a table big enough for the dense sweep to matter."""
        return bool(self.TWO_PASS_TABLE_ADAM and not self._dp and self._hip_read and not self.l2_reg and not self.compact_table_grad
                    and not self.lazy_table_adam and ids.shape[0] > 0 and self.flat_grad.numel() >= self.TWO_PASS_MIN_NUMEL
                    and self._table_adam_width_ok())

    def _train_step_two_pass(self, ids, label, keep_prob, masks, global_batch):
        """The same step with the dense table update (99.5 % of the parameters, 1.5 GB of HBM traffic, the longest kernel
        of the step's serial tail) split by what it depends on.  A row no id of the batch points at has an exactly-zero
        gradient: its update -- m = b1 m, v = b2 v, p -= lr_t m / (sqrt(v) + eps), the SAME arithmetic the dense sweep
        would do with g = 0 -- needs nothing this step computes, and nothing this step computes reads it (the gather and
        the scatter only touch the batch's rows).  So: mark the batch's rows, update all OTHER rows on the auxiliary
        stream underneath the forward scans (which leave HBM nearly idle), and behind the scatter update only the marked
        rows; that pass also clears the gradient rows and flags it consumed, so the table gradient is never cleared
        densely either.  Results are identical to the one-sweep path (tests/test_gpu_parity.py)."""
        n_emb = self.params["Embedding/emb_mtx"].numel()
        V, E = self.feature_size, self.embedding_size
        if self._row_flags is None:
            self._row_flags = torch.zeros(V, device=self.device, dtype=torch.uint8)
        flags = self._row_flags
        t = self.adam_t + 1
        lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        views = [b[:n_emb].view(V, E) for b in (self.flat_param, self.flat_grad, self.flat_m, self.flat_v)]

        def early():                                          # runs on the auxiliary stream
            if not self._table_grad_clean:
                self.flat_grad.zero_()                        # (a stand-alone compute_gradients left a gradient behind)
            else:
                self.flat_grad[n_emb:].zero_()

            def rest():                                       # (nothing of the forward or the read path waits for this)
                ops.table_mark_rows(ids, flags)
                v1 = V if self.EARLY_PASS_SPLIT >= 1.0 else max(1, min(V, int(V * self.EARLY_PASS_SPLIT)))
                probe = self._split_probe if isinstance(self._split_probe, dict) and self._split_probe.get("armed") else None
                if probe is not None:
                    probe["a0"].record()
                ops.adam_step_table(*[x[:v1] for x in views], flags[:v1], 0, lr_t, self.beta1, self.beta2, self.adam_eps,
                                    clip=1.0)
                if probe is not None:
                    probe["a1"].record()
                if v1 >= V:
                    return None

                def rest2():    # the other rows: behind the read launch (a pass longer than the forward runs into it: the
                    #             stream of 28 B per table element pushes the read path's weights out of the L2 it lives on)
                    ops.adam_step_table(*[x[v1:] for x in views], flags[v1:], 0, lr_t, self.beta1, self.beta2,
                                        self.adam_eps, clip=1.0)
                return rest2
            return rest

        self._tune_early_split()
        out, ce = self.compute_gradients(ids, label, keep_prob, masks, global_batch, defer_join=True, _clear_grads=early)
        pending = out.pop("pending", None)
        self.adam_t = t
        # (compute_gradients has ordered this stream behind the auxiliary one: pass 0 is complete)
        ops.adam_step_table(*views, flags, 1, lr_t, self.beta1, self.beta2, self.adam_eps, clip=1.0)
        self._table_grad_clean = True
        if pending is not None:
            pending.join()
        self.apply_gradients(n_emb, self.flat_param.numel(), advance=False)
        return out, ce

    # ------------------------------------------------------------------ the same step under data parallel
    ROWS_EXCHANGE_MIN_BYTES = int(os.environ.get("HPMN_ROWS_EXCHANGE_MIN_BYTES", str(1 << 30)))

    def _dp_two_pass(self, ids) -> bool:
        """world > 1 and everything _two_pass_table_adam asks for: the N-GPU step is then the 1-GPU step (two-pass
        dense table Adam, the four library calls) plus the collectives.  HPMN_TABLE_EXCHANGE: ``auto`` (default) =
        ``rows`` for tables above ROWS_EXCHANGE_MIN_BYTES, else ``allreduce``; ``sharded`` / ``single`` keep their
        one-sweep forms (train_step)."""
        return bool(self.TWO_PASS_TABLE_ADAM and self._dp and self._hip_read and not self.l2_reg and not self.compact_table_grad
                    and not self.lazy_table_adam and self.table_exchange in ("auto", "rows", "allreduce")
                    and self.flat_grad.numel() >= self.TWO_PASS_MIN_NUMEL and self._table_adam_width_ok())

    def _train_step_dp(self, ids, label, keep_prob, masks, global_batch):
        """_train_step_two_pass with the batch sharded over the ranks.  What a rank may treat as
        "untouched" is what NO rank touches, so the marking runs over all ranks' ids -- gathered at the start of the
        step (4 MB per rank at the reference batch, fixed size: no host synchronisation), underneath the forward like
        the early pass itself.  Behind the scatter the table gradient is exchanged either densely (all-reduce of the
        [V, E] gradient: 2 (N-1)/N x 212 MB per rank at C3) or as touched rows (all-gather of every rank's unique row
        ids + gradient rows, summed in rank order on every rank so the replicas stay bit-identical) -- then the late
        pass runs over the union's rows exactly as in the single-process step.

        r4: the ids are known before the step computes anything, so the unique row list and the ranks' counts are
        produced AT THE START of the step on the auxiliary stream (underneath the forward) and land in pinned host
        memory through an asynchronous copy; the host looks at them only when it sizes the row exchange, after it has
        enqueued forward + BPTT -- by then the copy finished milliseconds ago, so there is no host read on the critical
        path.  ``auto`` picks the exchange PER STEP from those counts by modelled bytes received per rank
        ((N-1) x max count x (4 + 4 E) for the rows against 2 (N-1)/N x table bytes for the ring all-reduce): the same
        numbers on every rank, hence the same decision."""
        n_emb = self.params["Embedding/emb_mtx"].numel()
        V, E = self.feature_size, self.embedding_size
        if global_batch is None:
            #  without the global batch the fixed-size id gather is sized from the LOCAL shard: every rank
            # must then hold the same number of rows -- ragged shards have to pass global_batch
            global_batch = ids.shape[0] * self.world
        if self._row_flags is None:
            self._row_flags = torch.zeros(V, device=self.device, dtype=torch.uint8)
        flags = self._row_flags
        t = self.adam_t + 1
        lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        views = [b[:n_emb].view(V, E) for b in (self.flat_param, self.flat_grad, self.flat_m, self.flat_v)]
        B = ids.shape[0]
        gb = global_batch
        per_sample = ids[0].numel() if B > 0 else self.spec.T * self.spec.F
        cap = max(dist.shard_sizes(gb, self.world)) * per_sample
        if B * per_sample > cap:
            raise ValueError("this rank's shard (%d rows) is larger than the largest shard of global_batch=%d over %d ranks"
                             % (B, gb, self.world))
        want_counts = self.table_exchange in ("auto", "rows")
        use_plan = bool(want_counts and self._det_env != "0" and self.embedding_size % 4 == 0
                        and 256 % (self.embedding_size // 4) == 0)
        self._plan_wants_rows = use_plan
        self.last_scatter_plan = None
        # The rows travel in C chunks of the table's row range (the plan's distinct rows are ascending: a chunk is a slice), so
        # that the sum + late table-Adam pass of chunk c run while RCCL is still gathering chunk c+1 -- the collective is
        # xGMI-bound, the pass HBM-bound.  Same addends in the same order per row: results do not depend on C.
        C = max(1, int(self.table_exchange_chunks)) if use_plan else 1
        bounds = [(V * k) // C for k in range(C + 1)]
        self._plan_row_bounds = bounds if use_plan else None
        box = {}

        def early():                                          # runs on the auxiliary stream
            if not self._table_grad_clean:
                self.flat_grad.zero_()
            else:
                self.flat_grad[n_emb:].zero_()

            def rest():
                all_ids = dist.gather_ids(ids, max(cap, 1))
                if want_counts:
                    plan = self.last_scatter_plan if (use_plan and B > 0) else None
                    if plan is not None:
                        # the scatter plan already holds this rank's distinct rows and their count ON THE DEVICE: no
                        # torch.unique (whose data-dependent output shape is a host synchronisation), no index_select later
                        box["plan"] = plan
                        torch.cuda.current_stream().wait_event(plan.ready)
                        box["counts"] = dist.exchange_counts_async(plan.chunk_counts, self.device)
                    elif use_plan:                                # (an empty shard: C zeros, like everybody's C counts)
                        box["counts"] = dist.exchange_counts_async(torch.zeros(C, device=self.device, dtype=torch.int64),
                                                                   self.device)
                    else:
                        rows = (torch.unique(ids.reshape(-1)).long() if B > 0
                                else torch.empty(0, device=self.device, dtype=torch.int64))
                        box["rows"] = rows
                        box["counts"] = dist.exchange_counts_async(rows.numel(), self.device)
                ops.table_mark_rows(all_ids, flags)
                ops.adam_step_table(*views, flags, 0, lr_t, self.beta1, self.beta2, self.adam_eps, clip=1.0)
            return rest

        if B > 0:
            out, ce = self.compute_gradients(ids, label, keep_prob, masks, gb, defer_join=True, _clear_grads=early)
        else:
            # an empty shard (short last batch): nothing to compute, every collective still has to be entered
            main = torch.cuda.current_stream()
            self._aux_stream.wait_stream(main)
            with torch.cuda.stream(self._aux_stream):
                early()()
            main.wait_stream(self._aux_stream)
            out, ce = dict(prediction=torch.empty(0, device=self.device)), torch.zeros((), device=self.device)
        pending = out.pop("pending", None)
        self.adam_t = t
        table_grad = views[1]
        mode = self.table_exchange
        counts = box["counts"].result() if want_counts else None
        if use_plan and want_counts:
            counts2d = [c if isinstance(c, list) else [c] for c in counts]     # [world][C] (C == 1 comes back flat)
            counts = [sum(c) for c in counts2d]                   # distinct rows per rank
        wide = V > 2 ** 31 - 1
        if mode == "auto":
            mode = ("rows" if dist.rows_exchange_bytes(counts, E, wide) < dist.dense_allreduce_bytes(n_emb, self.world)
                    else "allreduce")
        self.last_exchange_mode = mode
        late_done = False
        if mode == "rows" and use_plan:
            plan = box.get("plan")
            # every chunk's all-gather is started now (they queue on RCCL's stream in order); chunk c is consumed -- own rows
            # zeroed, every rank's rows added in rank order, late table-Adam pass over the chunk's row range -- while chunk
            # c+1 is still travelling
            inflight, p0 = [], 0
            for c in range(C):
                n_mine = counts2d[self.rank][c]
                if plan is not None:
                    rows_c, g_c = plan.rows[p0:p0 + n_mine].long(), plan.out_rows[p0:p0 + n_mine]
                else:
                    rows_c = torch.empty(0, device=self.device, dtype=torch.int64)
                    g_c = torch.empty(0, E, device=self.device, dtype=torch.float32)
                p0 += n_mine
                cc = [counts2d[r][c] for r in range(len(counts2d))]
                inflight.append((rows_c, cc) + dist.exchange_rows(rows_c, g_c, cc, wide_ids=wide, async_op=True))
            self.last_exchange_bytes = sum(dist.rows_exchange_bytes(x[1], E, wide) for x in inflight)
            for c, (rows_c, cc, ids_all, g_all, works) in enumerate(inflight):
                for w in works:
                    w.wait()                                      # (orders the current stream behind the collective)
                table_grad.index_fill_(0, rows_c, 0.0)            # (own rows come back through g_all, in rank order)
                dist.sum_rows_into_(table_grad, ids_all, g_all, cc)
                v0, v1 = bounds[c], bounds[c + 1]
                if v1 > v0:
                    ops.adam_step_table(*[x[v0:v1] for x in views], flags[v0:v1], 1, lr_t, self.beta1, self.beta2,
                                        self.adam_eps, clip=1.0)
            late_done = True
        elif mode == "rows":
            rows = box["rows"]
            rows.record_stream(torch.cuda.current_stream())   # (made on the auxiliary stream, consumed here)
            mine = table_grad.index_select(0, rows)
            ids_all, g_all = dist.exchange_rows(rows, mine, counts, wide_ids=wide)
            table_grad.index_fill_(0, rows, 0.0)              # (own rows come back through g_all, in rank order)
            dist.sum_rows_into_(table_grad, ids_all, g_all, counts)
            self.last_exchange_bytes = dist.rows_exchange_bytes(counts, E, wide)
        else:
            dist.allreduce_sum_(self.flat_grad[:n_emb])
            self.last_exchange_bytes = dist.dense_allreduce_bytes(n_emb, self.world)
        if not late_done:
            ops.adam_step_table(*views, flags, 1, lr_t, self.beta1, self.beta2, self.adam_eps, clip=1.0)
        self._table_grad_clean = True
        if pending is not None:
            pending.join()
        dist.allreduce_sum_(self.flat_grad[n_emb:])
        self.apply_gradients(n_emb, self.flat_param.numel(), advance=False)
        return out, ce

    # ------------------------------------------------------------------ r5: the step without a dense gradient table
    def _rows_geometry(self, ids, global_batch):
        """What the plan of a step's scatter needs beyond the ids: data parallel -> (global batch, capacity of the row list
        = the largest shard's lookups, chunk count, row boundaries of the chunks); one process -> (global_batch, 0, 0, None)."""
        if not self._dp:
            return global_batch, 0, 0, None
        B = ids.shape[0]
        gb = B * self.world if global_batch is None else global_batch      # (ragged shards must pass it)
        per_sample = ids[0].numel() if B > 0 else self.spec.T * self.spec.F
        cap = max(1, max(dist.shard_sizes(gb, self.world)) * per_sample)
        if B * per_sample > cap:
            raise ValueError("this rank's shard (%d rows) is larger than the largest shard of global_batch=%d over %d ranks"
                             % (B, gb, self.world))
        C = max(1, min(int(self.table_exchange_chunks), ops._lib.HPMN_MAX_CHUNKS))
        V = self.feature_size
        return gb, cap, C, [(V * k) // C for k in range(C + 1)]

    def _rows_plan(self, ids, cap, bounds):
        """The scatter plan of ``ids`` on the auxiliary stream (the CURRENT stream when called)."""
        pl = ops.ScatterPlan(ids, self.embedding_size, want_rows=True, row_bounds=bounds,
                             rows_capacity=cap, out_rows_capacity=(ids.numel() + cap) if cap else 0, V=self.feature_size)
        pl.ready = torch.cuda.Event()
        pl.ready.record()
        return pl

    def _rows_early_exchange(self, plan, ids_dtype, cap, C, group=None):
        """Data parallel, on the current stream: every rank's distinct-row list and counts, as they are."""
        if plan is not None:
            rows_mine, cnt = plan.rows, plan.counts_vec         # [cap]; [distinct rows, and per chunk of the row range]
        else:                                                 # (an empty shard: nothing to send, every collective entered)
            rows_mine = torch.empty(cap, device=self.device, dtype=ids_dtype)
            cnt = torch.zeros(1 + C, device=self.device, dtype=torch.int32)
        ids_all = dist.all_gather_fixed(rows_mine, group=group)            # [world, cap]
        cnt_all = dist.all_gather_fixed(cnt, group=group)                  # [world, 1 + C] int32
        return dict(ids_all=ids_all, cnt_all=cnt_all, counts=dist.HostCopy(cnt_all))

    def _prefetch_rows(self, next_ids, next_global_batch, after=None):
        """train_step(next_ids=): the NEXT step's plan (and, data parallel, its early exchange -- on a second communicator, so
        that it never queues in front of this step's collectives) on a stream of its own, enqueued in front of this step: it
        has the whole step to finish in, on whatever the scans leave free, and nothing of this step waits for it.
        (r5, measured: underneath layer 0's reverse scan it took the weight gradients' slots -- the step's tail grew by 150 us --
        and its counts reached the host so late that the next step started 0.3 ms behind an idle device.)"""
        if self._plan_stream is None:
            # (plain priority: with the plan's ~20 sort launches at high priority the rows step went from 2.6 to 4.3 ms -- they are
            #  dispatched in front of the scans' workgroups.  HPMN_PLAN_ON_AUX=1: no stream of its own -- the auxiliary stream,
            #  behind the early table pass: one stream fewer to share the runtime's hardware queues)
            self._plan_stream = (self._aux_stream if os.environ.get("HPMN_PLAN_ON_AUX", "0") == "1"
                                 else torch.cuda.Stream(device=self.device))
        pst = self._plan_stream
        gb, cap, C, bounds = self._rows_geometry(next_ids, next_global_batch)
        # The early exchange goes over a SECOND communicator (HPMN_DP_SIDE_GROUP=0: the default one).  Measured with one rank on
        # RCCL in bench.py's timed loop: 2.60 ms/step against 2.97 on the one communicator -- its collectives then sit on that
        # communicator's stream in front of the running step's.  Ordering argument for N > 1: a rank's side-communicator calls
        # for step N+2 are enqueued after its main-communicator calls of step N and before those of step N+1, on every rank
        # alike; neither kind waits for the other on the device (compute kernels never wait for a collective), so a peer that
        # lags finishes its main-communicator calls first and then joins -- no cycle.  Unmeasured on more than one GPU.
        # r6: the second communicator is OPT-IN (HPMN_DP_SIDE_GROUP=1) until a run on more than one GPU has validated
        # it -- two communicators' kernels in flight on streams that may share a hardware queue (seven streams, four or five
        # queues) is a known RCCL hang hazard: rank A's queue could hold [side, main] where rank B's holds [main, side].  On the
        # default communicator every rank issues every collective in the same program order.
        group = dist.side_group() if (self._dp and os.environ.get("HPMN_DP_SIDE_GROUP", "0") == "1") else None
        # (the plan stream reads next_ids -- order it behind what the caller's stream has been given so far, in case
        #  the tensor was produced there; that is the END of the previous step, the plan still has this whole step to run in.
        #  next_ids must stay unchanged until the next train_step has consumed the plan: a slice of a staged dataset does.)
        if after is not None:
            pst.wait_event(after)                             # (the caller's stream as it was when the step STARTED)
        else:
            pst.wait_stream(torch.cuda.current_stream())
        # r6: WITHOUT the second communicator the early exchange is issued at the END of this step (_prefetch_exchange), not here:
        # collectives of one communicator run in issue order on its stream, and issued here they sat in front of THIS step's row
        # exchange and dense all-reduce waiting for a plan that is built on whatever the scans leave free (one rank on RCCL, C3:
        # 3.35 ms/step against 3.21 with no hint at all and 2.95 on the second communicator).  At the end of the step the plan
        # is long done and nothing of this step queues behind it.
        defer = bool(self._dp and group is None)
        with torch.cuda.stream(pst):
            plan = self._rows_plan(next_ids, cap, bounds) if next_ids.shape[0] > 0 else None
            ex = self._rows_early_exchange(plan, next_ids.dtype, cap, C, group=group) if (self._dp and not defer) else None
            done = torch.cuda.Event()
            done.record()
        if plan is not None:
            plan.ready = done
        self._prefetched = dict(key=(next_ids.data_ptr(), tuple(next_ids.shape), gb), plan=plan, ex=ex, ids=next_ids, done=done,
                                defer=(next_ids.dtype, cap, C) if defer else None)

    def _prefetch_exchange(self):
        """End of a data-parallel rows step: the NEXT step's early exchange (every rank's distinct-row list and counts) on the
        default communicator, behind this step's collectives; the plan stream carries it (its plan is there)."""
        pf = self._prefetched
        if pf is None or pf.get("defer") is None:
            return
        dtype, cap, C = pf["defer"]
        pst = self._plan_stream
        with torch.cuda.stream(pst):
            pf["ex"] = self._rows_early_exchange(pf["plan"], dtype, cap, C, group=None)
            done = torch.cuda.Event()
            done.record()
        pf["done"], pf["defer"] = done, None
        if pf["plan"] is not None:
            pf["plan"].ready = done

    # MEASUREMENT ONLY: HPMN_DP_WIRE_STANDIN="<GB/s>,<ranks>[,<fraction of the rows>]" -- with ONE rank on RCCL the
    # all-gathers are local copies; a stream of its own then holds every chunk back by the time the bytes this rank would
    # RECEIVE at <ranks> ranks take at <GB/s> (sleep kernels, one behind the other like transfers on one link, each starting
    # when its chunk's rows exist), and the late launches wait for that instead of for the copy.  What the step gains over the
    # run without it is the exchange's EXPOSED time with this step's real kernels around it.
    WIRE_STANDIN = os.environ.get("HPMN_DP_WIRE_STANDIN", "")
    _wire_stream = None
    _sleep_cycles_per_us = None

    def _wire_standin(self, windows, inflight, E):
        gbps, ranks, frac = (list(map(float, self.WIRE_STANDIN.split(","))) + [1.0])[:3]
        if self._wire_stream is None:
            self._wire_stream = torch.cuda.Stream(device=self.device)
            # torch.cuda._sleep counts ticks of a device clock: calibrate once
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); torch.cuda._sleep(20_000_000); b.record(); b.synchronize()
            self._sleep_cycles_per_us = 20_000_000 / (a.elapsed_time(b) * 1e3)
        out = []
        with torch.cuda.stream(self._wire_stream):
            for (first, n, capc), item in zip(windows, inflight):
                if item is None:
                    out.append(None)
                    continue
                if item[1] is not None:
                    item[1].wait()                            # (this stream behind the chunk's collective)
                us = (ranks - 1.0) * frac * capc * E * 4 / (gbps * 1e3)
                torch.cuda._sleep(max(1, int(us * self._sleep_cycles_per_us)))
                ev = torch.cuda.Event()
                ev.record()
                item[0].record_stream(torch.cuda.current_stream())
                out.append(ev)
        return out

    def _train_step_rows(self, ids, label, keep_prob, masks, global_batch, next_ids=None, next_global_batch=None):
        """The two-pass step on COMPACT gradient rows, one process or N (``compact_table_grad``).

        Start of the step, auxiliary stream, underneath the forward: the plan of the deterministic scatter (a stable sort of
        the batch's ids: distinct rows ascending, their count on the device); the marking of the rows SOMEBODY touches; the
        early pass of the table Adam over all other rows (gradient exactly zero).  Data parallel: the plan's row buffer (its
        capacity comes from the batch geometry, the same on every rank) and a vector of 1 + C counts are all-gathered there
        as they are, and the marking sets one bit per rank from the gathered lists -- 2 collectives and 2 kernels, no padding
        copies; the counts travel to pinned memory for the moment the host sizes the row exchange.
        Behind BPTT: the scatter's segmented reduction writes the compact gradient rows (no dense table), and
        ``hpmn_rows_sum_adam`` updates the touched rows -- one launch; data parallel: the rows go out as C slices of the plan's own
        buffer (C all-gathers started back to back), and per chunk one launch adds every rank's rows in rank order, clips and
        applies Adam while the next chunk is still on the wire.  The dense variables follow as before (join, all-reduce,
        one small Adam launch)."""
        V, E = self.feature_size, self.embedding_size
        n_emb = V * E
        if self._row_flags is None:
            self._row_flags = ops.table_flags(V, self.device)
        flags = self._row_flags
        t = self.adam_t + 1
        lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        hp = dict(beta1=self.beta1, beta2=self.beta2, eps=self.adam_eps, clip=1.0)
        P, M, S = (b[:n_emb].view(V, E) for b in (self.flat_param, self.flat_m, self.flat_v))
        B = ids.shape[0]
        dp = self._dp
        box = {}
        gb, cap, C, bounds = self._rows_geometry(ids, global_batch)
        self._plan_wants_rows, self._plan_row_bounds = bool(dp), bounds
        self._plan_caps = (cap, B * (ids[0].numel() if B > 0 else 0) + cap) if dp else (0, 0)
        self.last_scatter_plan = None
        # what the previous call prepared for this one (train_step(next_ids=)): the plan and, data parallel, the exchanged lists
        pre, self._prefetched = self._prefetched, None
        if pre is not None and pre["key"] != (ids.data_ptr(), tuple(ids.shape), gb):
            pre = None        # (not the batch that was announced: prepared for nothing.  The same on every rank: the key is a
            #                    function of the call sequence, which data parallel requires to be the same everywhere)
        if pre is not None:
            # (made on the plan stream: the consumers' streams wait for its event and hold its buffers)
            self._aux_stream.wait_event(pre["done"])
            if pre["plan"] is not None:
                pre["plan"].record_stream(torch.cuda.current_stream())
                pre["plan"].record_stream(self._aux_stream)
            self._preset_plan = pre["plan"]
            if pre["ex"] is not None:
                for x in (pre["ex"]["ids_all"], pre["ex"]["cnt_all"]):
                    x.record_stream(self._aux_stream)
                box.update(pre["ex"])
        step_start = None
        if next_ids is not None:
            step_start = torch.cuda.Event()
            step_start.record()
        probe = self._phase_probe
        if probe is not None:
            pe = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            pe[0].record()
        # (r6: the NEXT step's plan is enqueued BEHIND this step's forward + BPTT, not in front of them: under data parallel the
        #  host waits for the exchanged counts once per step, so it is never far ahead of the device, and the plan's ~20 launches
        #  in front of the forward were 0.3 ms of every step during which the device waited for the host -- timeline
        #  profiles/r06_timeline_rows.txt: the forward started 365 us into the step)

        def early():                                          # runs on the auxiliary stream
            self.flat_grad.zero_()                            # (the dense variables' gradient: a few hundred kB)

            def rest():
                if not dp:
                    if B > 0:
                        ops.table_mark_rows(ids, flags)
                else:
                    if "ids_all" not in box:                  # (not prepared underneath the previous step)
                        plan = self.last_scatter_plan if B > 0 else None
                        if plan is not None:
                            torch.cuda.current_stream().wait_event(plan.ready)
                        box.update(self._rows_early_exchange(plan, ids.dtype, cap, C))
                    if self.world > 1:
                        # (a bucket index over the gathered lists, built by the marking pass: the late launch's searches then
                        #  probe a handful of entries instead of a whole list)
                        box["buckets"] = ops.RowBuckets(self.world, V, cap, self.device)
                        box["buckets"].start.record_stream(torch.cuda.current_stream())
                    ops.table_mark_ranks(box["ids_all"], box["cnt_all"], flags, counts_stride=1 + C, buckets=box.get("buckets"))
                v1 = V if self.EARLY_PASS_SPLIT >= 1.0 else max(1, min(V, int(V * self.EARLY_PASS_SPLIT)))
                probe = self._split_probe if isinstance(self._split_probe, dict) and self._split_probe.get("armed") else None
                if probe is not None:
                    probe["a0"].record()
                ops.adam_step_table(P[:v1], None, M[:v1], S[:v1], flags[:v1], 0, lr_t, **hp)
                if probe is not None:
                    probe["a1"].record()
                if v1 >= V:
                    return None

                def rest2():                                  # (the rows the forward left no room for: beside BPTT)
                    ops.adam_step_table(P[v1:], None, M[v1:], S[v1:], flags[v1:], 0, lr_t, **hp)
                return rest2
            return rest

        self._tune_early_split()
        if B > 0:
            out, ce = self.compute_gradients(ids, label, keep_prob, masks, gb, defer_join=True, _clear_grads=early)
        else:
            main = torch.cuda.current_stream()
            self._aux_stream.wait_stream(main)
            with torch.cuda.stream(self._aux_stream):
                r2 = early()()
                if callable(r2):
                    r2()
            main.wait_stream(self._aux_stream)
            out, ce = dict(prediction=torch.empty(0, device=self.device)), torch.zeros((), device=self.device)
        pending = out.pop("pending", None)
        self.adam_t = t
        plan = self.last_scatter_plan if B > 0 else None
        self._preset_plan = None
        if probe is not None:
            pe[1].record()                                    # (forward + read path + BPTT + the scatter's segmented reduction)
        if next_ids is not None:
            # (its own stream, ordered behind the caller's stream as it was at the START of the step -- not behind the BPTT just
            #  enqueued: it runs on whatever this step's scans leave free)
            self._prefetch_rows(next_ids, next_global_batch, after=step_start)
        if not dp:
            if plan is not None:
                ops.rows_sum_adam(P, M, S, flags, plan.rows.view(1, -1), plan.out_rows.view(1, -1, E), lr_t,
                                  counts=plan.count, **hp)
        else:
            main = torch.cuda.current_stream()
            ids_all, cnt_all = box["ids_all"], box["cnt_all"]
            for x in (ids_all, cnt_all) + ((box["buckets"].start,) if "buckets" in box else ()):
                x.record_stream(main)                         # (made on the auxiliary stream, consumed here)
            lens, windows = dist.rows_windows(box["counts"].result())      # (an event wait, long satisfied)
            src = plan.out_rows if plan is not None else torch.empty(cap, E, device=self.device, dtype=torch.float32)
            wide = ids.dtype == torch.int64
            # every chunk's all-gather is started now (they queue on RCCL's stream in order); chunk c is consumed -- all
            # ranks' rows added in rank order, clip, Adam, in one launch -- while chunk c + 1 is still travelling
            inflight = []
            for first, n, capc in windows:
                a = first[self.rank]
                inflight.append(dist.all_gather_fixed(src[a:a + capc], async_op=True) if capc > 0 else None)
            wired = self._wire_standin(windows, inflight, E) if self.WIRE_STANDIN else None
            for ci, ((first, n, capc), item) in enumerate(zip(windows, inflight)):
                if item is None:
                    continue
                g_all, work = item
                if wired is not None:
                    main.wait_event(wired[ci])                # (the chunk "arrives" when its stand-in transfer ends)
                elif work is not None:
                    work.wait()                               # (orders the current stream behind the collective)
                ops.rows_sum_adam(P, M, S, flags, ids_all, g_all, lr_t, lens=lens, first=first, n=n,
                                  buckets=box.get("buckets"), **hp)
            self.last_exchange_mode = "rows"
            self.last_exchange_bytes = dist.rows_exchange_bytes_windows(windows, E, wide, self.world, cap)
        if probe is not None:
            pe[2].record()                                    # (the rows exchange and the touched rows' update)
        if pending is not None:
            pending.join()
        lo = self._goff
        dist.allreduce_sum_(self.flat_grad)                   # (the flat gradient holds the dense variables only)
        ops.adam_step(self.flat_param[lo:], self.flat_grad, self.flat_m[lo:], self.flat_v[lo:], lr_t, self.beta1,
                      self.beta2, self.adam_eps, clip=1.0)
        if probe is not None:
            pe[3].record()                                    # (join of layer 0's weight gradient, dense all-reduce, dense Adam)
            probe.append(pe)
        if dp:
            self._prefetch_exchange()                         # (the next step's lists and counts, behind this step's collectives)
        return out, ce

    def table_gradient(self) -> torch.Tensor:
        """The embedding-table gradient of the last compute_gradients as a dense [V, E] tensor (inspection / tests): the
        flat buffer's view, or -- ``compact_table_grad`` -- the scatter plan's compact rows spread out."""
        if not self.compact_table_grad:
            return self.grads["Embedding/emb_mtx"]
        g = torch.zeros(self.feature_size, self.embedding_size, device=self.device, dtype=torch.float32)
        plan = self.last_scatter_plan
        if plan is not None:
            u = int(plan.count.item())
            g[plan.rows[:u].long()] = plan.out_rows[:u]
        return g

    def apply_gradients(self, lo: int = 0, hi: Optional[int] = None, advance: bool = True, clear: bool = False):
        """clip + TF-form Adam over elements [lo, hi) of the flat buffers (default: everything);
        ``advance`` = this call starts a new optimiser step; ``clear``: the gradient elements are consumed (left zero)."""
        if self.compact_table_grad:
            raise RuntimeError("compact_table_grad: there is no dense table gradient to sweep -- use train_step "
                               "(HPMN_TABLE_GRAD=dense restores the flat gradient over the table)")
        if advance:
            self.adam_t += 1
        t = self.adam_t
        lr_t = self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)
        hi = self.flat_param.numel() if hi is None else hi
        with torch.no_grad():
            ops.adam_step(self.flat_param[lo:hi], self.flat_grad[lo:hi], self.flat_m[lo:hi], self.flat_v[lo:hi],
                          lr_t, self.beta1, self.beta2, self.adam_eps, clip=1.0, clear_grad=clear)

    # ------------------------------------------------------------------ datasets
    def _dev(self, dataset) -> _DeviceDataset:
        """Device-resident copy of ``dataset``, staged once per object.  The cache entry keeps a reference to the
        dataset it was built from and is only used for that very object (an ``id`` can be recycled once a
        temporary dataset is collected).  A list mutated in place after staging (e.g. shuffled between epochs)
        must be re-staged with ``invalidate_dataset``; at most ``max_cached_datasets`` stay resident."""
        key = id(dataset)
        hit = self._datasets.get(key)
        if hit is not None and hit[0] is dataset:
            return hit[1]
        ds = _DeviceDataset(dataset, self.device, self.industry, getattr(self, "feature_size", None),
                            want_item=bool(getattr(self, "item", False)))
        while len(self._datasets) >= self.max_cached_datasets:
            # evict the oldest entry that is not the model's own train / test set
            victim = next((k for k, (d, _) in self._datasets.items()
                           if d is not self.trainset and d is not self.testset), None)
            if victim is None:
                break
            del self._datasets[victim]
        self._datasets[key] = (dataset, ds)
        return ds

    max_cached_datasets = 8

    def invalidate_dataset(self, dataset=None):
        """Drop the device copy of ``dataset`` (all copies when None): the next use re-stages it."""
        if dataset is None:
            self._datasets.clear()
        else:
            self._datasets.pop(id(dataset), None)

    # ------------------------------------------------------------------ harness (code/hpmn.py:467-519)
    def train(self, epochs, batchsize):
        step, count, best = 0, 0, 0.0
        ds = self._dev(self.trainset)
        self._prefetched = None
        order = [bh for _ in range(epochs) for bh in ds.batches(batchsize)]      # stored order, every epoch (code/hpmn.py:470-472)
        for k, (lo, hi) in enumerate(order):
            step += 1
            # data parallel: every rank takes a contiguous slice of the global batch
            a, b = dist.shard_bounds(lo, hi, self.rank, self.world)
            nxt = {}
            if self.compact_table_grad and k + 1 < len(order):
                # the next batch's ids are known: its scatter plan / row exchange is prepared underneath this step's BPTT
                lo2, hi2 = order[k + 1]
                a2, b2 = dist.shard_bounds(lo2, hi2, self.rank, self.world)
                nxt = dict(next_ids=ds.ids[a2:b2], next_global_batch=hi2 - lo2)
            self.train_step(ds.ids[a:b], ds.label[a:b], keep_prob=0.5, global_batch=hi - lo,
                            item_ids=None if ds.item_ids is None else ds.item_ids[a:b], **nxt)
            if step % self.eval_every == 0:
                result = list(self.eval(self.trainset, 4 * batchsize))
                result += list(self.eval(self.testset, 4 * batchsize))
                self.log(step, result)
                if not (result[3] > best):
                    count += 1
                    if count > 3:
                        return best
                else:
                    count = 0
                    best = result[3]
        return best

    def eval(self, dataset, batchsize):
        """-> (auc, log-loss, mean of per-batch memory_loss); code/hpmn.py:497-519.  Data parallel (r6): the
        DATASET is sharded -- rank r scores rows [n r / N, n (r+1) / N) in full-width passes on the same kernels as the single
        process -- and ONE all-gather of the predictions plus ONE all-reduce of the memory-loss sum end the pass (r5: every
        reference batch was cut into per-rank slices, two collectives per batch, 250-row slices below the tile kernels' width)."""
        ds = self._dev(dataset)
        preds, mem_losses = [], []
        user_only = ds.item_ids is None and getattr(self, "_hip_read", False)
        if user_only and (self._dp or (ds.n > batchsize and self._tiled_inference(self.TILED_EVAL_ROWS))):
            # User-only graph: SEVERAL reference batches per pass (the tile kernel wants ~4096 rows in
            # flight; the harness's 4 x 500 = 2000 half-fill the chip).  The reference's third return value is the mean over
            # batches of the per-batch memory_loss SUMS (code/hpmn.py:360-369, :512-519) = the sum over all rows divided by the
            # number of reference batches, whatever the grouping -- so only the float32 summation order differs.
            n_ref = -(-ds.n // batchsize)
            # (<= 4096 rows = 256 tiles: one round of CUs on the kernels that hold a CU per tile; H = 64: the four-wave kernel
            #  takes two tiles per CU -- 8192 rows per pass, r5)
            rows_per_pass = self.TILED_EVAL_ROWS * (2 if (self.spec.H == 64 and ops.TILE64 and self.TILED_EVAL_ROWS == 4096) else 1)
            per_pass = batchsize * max(1, rows_per_pass // batchsize)
            total = torch.zeros(1, device=self.device)
            first, last = dist.shard_bounds(0, ds.n, self.rank, self.world)      # (single process: the whole set)
            for lo in range(first, last, per_pass):
                out = self.forward_inference(ds.ids[lo:min(lo + per_pass, last)], want_logit=False, want_att=False)
                preds.append(out["prediction"])
                total += out["memory_loss"].reshape(1)
            if self._dp:
                mine = torch.cat(preds) if preds else torch.empty(0, device=self.device)
                preds = [dist.gather_predictions(mine.contiguous(), ds.n)]
                dist.allreduce_sum_(total)
            mem_losses = [total / float(n_ref)]
            batches = ()
        else:
            batches = ds.batches(batchsize)
        for lo, hi in batches:
            a, b = dist.shard_bounds(lo, hi, self.rank, self.world)
            out = self.forward_inference(ds.ids[a:b], item_ids=None if ds.item_ids is None else ds.item_ids[a:b])
            pred, ml = out["prediction"], out["memory_loss"].reshape(1)
            if self._dp:
                pred = dist.gather_predictions(pred.contiguous(), hi - lo)
                dist.allreduce_sum_(ml)
            preds.append(pred)
            mem_losses.append(ml)
        # metrics on the device (the whole pass is enqueued without a host round trip; ONE sync for three
        # scalars): same definitions as sklearn's roc_auc_score / log_loss used at code/hpmn.py:516-518
        preds = torch.cat(preds).double()
        auc = device_auc(preds, ds.label)
        loss = device_log_loss(preds, ds.label)
        mem_loss = torch.cat(mem_losses).mean().double()
        auc, loss, mem_loss = torch.stack([auc, loss, mem_loss]).tolist()
        if math.isnan(auc):
            # sklearn.metrics.roc_auc_score (code/hpmn.py:516) raises this; a NaN would also poison train()'s
            # early-stopping comparisons silently
            raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
        return auc, loss, mem_loss

    def _first_hop_weights(self, ds, lo, hi):
        """self.user_weights of code/hpmn.py:534-537 (the item branch's weights when only that branch runs)."""
        out = self.forward_inference(ds.ids[lo:hi], item_ids=None if ds.item_ids is None else ds.item_ids[lo:hi])
        w = out["user_weights"] if out.get("user_weights") is not None else out["item_weights"]
        return w.cpu().numpy()

    def get_weights(self):
        """code/hpmn.py:521-560: first-hop attention weights over train+test at batch 512."""
        if self.rank != 0:          # unsharded forward + file writes: one rank does it (like save_model / log)
            return
        weights, lengths, labels = [], [], []
        for dataset in (self.trainset, self.testset):
            ds = self._dev(dataset)
            for lo, hi in ds.batches(512):
                weights.append(self._first_hop_weights(ds, lo, hi))
            if ds.length_np is not None:
                lengths.append(ds.length_np)
            labels.append(ds.label_np)
        np.save(self._path + "/weights.npy", np.concatenate(weights))
        if lengths:
            np.save(self._path + "/lengths.npy", np.concatenate(lengths))
        np.save(self._path + "/labels.npy", np.concatenate(labels))

    # ------------------------------------------------------------------ artefacts
    @property
    def save_path(self):
        if self._save_path is None:
            d = "%s/ckpt" % self._path
            os.makedirs(d, exist_ok=True)
            self._save_path = os.path.join(d, "model.ckpt")
        return self._save_path

    def save_model(self, global_step=None):
        """code/hpmn.py:91-92: ``saver.save(sess, save_path)`` -- written as the TensorFlow-1 tensor bundle the
        reference's Saver produces (``model.ckpt.index`` + ``model.ckpt.data-00000-of-00001`` + ``checkpoint``),
        under the variable names of the TF 1.4 graph, Adam slots and beta powers included
        (hpmn_amd/tf_checkpoint.py).  This side restores either implementation's checkpoint.  The reference's
        ``Saver().restore`` of ours needs a ``var_list`` limited to the exported names when the model is user-only: its
        graph also creates the (never executed) item-branch variables and their Adam slots (code/hpmn.py:444-450),
        which a user-only export does not contain.
        Every rank must call this under data parallel with the sharded table exchange: each rank holds the Adam moments
        of its own 1/world of the table rows only, and they are gathered here before rank 0 writes."""
        if self._dp and self._sharded_moments:
            # (the sharded update ran: every rank holds 1/world of the table's moments -- a collective, so EVERY rank has
            # to call save_model then; README / INTEGRATION state it.  Keyed on what ran, not on the switch: ADVICE r3)
            n_pad = self._emb_numel_padded
            shard = n_pad // self.world
            for buf in (self.flat_m, self.flat_v):
                dist.all_gather_shards_(buf[:n_pad], self.rank * shard, shard)
        if self.rank != 0:
            return
        from . import tf_checkpoint as tfc
        path = self.save_path if global_step is None else "%s-%s" % (self.save_path, global_step)
        torch.cuda.synchronize(self.device)
        host = lambda flat: {k: flat[self._offs[k]:self._offs[k] + v.numel()].view(v.shape).cpu().numpy()
                             for k, v in self.params.items()}
        tfc.export_model(path, host(self.flat_param), host(self.flat_m), host(self.flat_v), self.adam_t, self.beta1,
                         self.beta2, mask_table_rows=self.feature_size if self.spec.mask_id0 else None)

    def load_model(self, path: Optional[str] = None):
        """code/hpmn.py:105-111: ``saver.restore``; raises IOError like the reference when it fails.  Restores a
        checkpoint written by ``save_model`` or by the reference's TF 1.4 Saver (variables the User branch does not
        have -- the never-executed item branch -- are ignored; Adam state is taken when present)."""
        from . import tf_checkpoint as tfc
        path = self.save_path if path is None else path
        try:
            params, m, v, t = tfc.import_model(path, {k: tuple(p.shape) for k, p in self.params.items()}, self.beta1,
                                               self.beta2)
            self.set_params(params)
            with torch.no_grad():
                if m is not None:
                    for k in params:
                        o, n = self._offs[k], self.params[k].numel()
                        self.flat_m[o:o + n].copy_(torch.as_tensor(m[k]).reshape(-1))
                        self.flat_v[o:o + n].copy_(torch.as_tensor(v[k]).reshape(-1))
                else:
                    self.flat_m.zero_()
                    self.flat_v.zero_()
            self.adam_t = int(t)
        except Exception:
            raise IOError("Failed to load model from save path: %s" % path)
        print("Successfully load model from save path: %s" % path)

    def log(self, step, result):
        if self.rank != 0:
            return
        if self.verbose:
            print("Step: %s\tTrain AUC: %.5f\tTrain Loss: %.5f\tTrain Mem_loss: %.5f"
                  "\tTest AUC: %.5f\tTest Loss: %.5f\tTest Mem_loss: %.5f" % ((str(step),) + tuple(result[:6])))
        with open(self._path + "/result.log", "a") as fout:
            fout.write("%s\t%.5f\t%.5f\t%.5f\t%.5f\t%.5f\t%.5f\n" % ((str(step),) + tuple(result[:6])))


class Hpmn_Industry(Hpmn_Basic):
    """XLong graph (code/hpmn.py:217-410): bare embedding lookup, 23 zero steps in front
    (1001 -> 1024), query row = position -2, eval every 10 steps."""

    eval_every = 10
    industry = True

    def __init__(self, path, trainset, testset, feature_size, user_dim, item_dim, user_maxlen, item_maxlen,
                 learning_rate, hidden_size, embedding_size, hop, user_layers, item_layers, user_num_layers,
                 item_num_layers, user, item, emb_initializer=None, l2_reg=0, memory_reg=1e-5, **kw):
        self.user_maxlen, self.item_maxlen = int(user_maxlen), int(item_maxlen)
        super(Hpmn_Industry, self).__init__(path, trainset, testset, feature_size, user_dim, item_dim,
                                            learning_rate, hidden_size, embedding_size, hop, user_layers,
                                            item_layers, user_num_layers, item_num_layers, user, item,
                                            emb_initializer, l2_reg, memory_reg, **kw)

    def _make_spec(self) -> ScanSpec:
        # code/hpmn.py:288-292
        return ScanSpec(F=self.user_dim, E=self.embedding_size, H=self.hidden_size, K=self.user_num_layers,
                        T=self.user_maxlen, periods=tuple(self.user_layers[:self.user_num_layers]),
                        front_zero=23,     # literal of code/hpmn.py:288 (1001 + 23 = 1024 = the literal of :290)
                        mask_id0=False, last_index=-2)

    def _make_item_spec(self) -> ScanSpec:
        # code/hpmn.py:297-302: 192 - 184 = 8 zero steps in front, build_memory(iinp, item_layers, 192, ...),
        # last = iinp[:, -1, :]
        return ScanSpec(F=self.item_dim, E=self.embedding_size, H=self.hidden_size, K=self.item_num_layers,
                        T=self.item_maxlen, periods=tuple(self.item_layers[:self.item_num_layers]),
                        front_zero=192 - 184, mask_id0=False, last_index=-1)

    def get_weights(self):
        """code/hpmn.py:375-410."""
        if self.rank != 0:
            return
        weights, ids = [], []
        for dataset in (self.trainset, self.testset):
            ds = self._dev(dataset)
            for lo, hi in ds.batches(512):
                weights.append(self._first_hop_weights(ds, lo, hi))
                ids.append(ds.ids[lo:hi, :, 1].cpu().numpy())
        np.save(self._path + "/weights_new.npy", np.concatenate(weights))
        np.save(self._path + "/ids.npy", np.concatenate(ids))


class Hpmn(Hpmn_Industry):
    """Amazon/Taobao graph (code/hpmn.py:413-560): masked embedding (id 0 -> zero row), no zero
    prefix, query row = the target (position -1), eval every 100 steps."""

    eval_every = 100
    industry = False

    def _make_spec(self) -> ScanSpec:
        return ScanSpec(F=self.user_dim, E=self.embedding_size, H=self.hidden_size, K=self.user_num_layers,
                        T=self.user_maxlen, periods=tuple(self.user_layers[:self.user_num_layers]),
                        front_zero=0, mask_id0=True, last_index=-1)

    def _make_item_spec(self) -> ScanSpec:
        # code/hpmn.py:444-447 (the same masked embedding, :424-428)
        return ScanSpec(F=self.item_dim, E=self.embedding_size, H=self.hidden_size, K=self.item_num_layers,
                        T=self.item_maxlen, periods=tuple(self.item_layers[:self.item_num_layers]),
                        front_zero=0, mask_id0=True, last_index=-1)

    get_weights = Hpmn_Basic.get_weights


# ---------------------------------------------------------------------------------------
# CLI (code/hpmn.py:563-667): same argv, same relative data paths, same hyper-parameters
# ---------------------------------------------------------------------------------------
def main(argv: Sequence[str]) -> int:
    from .preprocess import load_dataset as load_dataset_pkl     # the pickle, through its int32 array cache
    if len(argv) != 2:
        print("Useage: python hpmn.py [dataset]")
        return 1
    dataset_name = argv[1]
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
        # `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 hpmn.py <dataset>`: one rank per GPU,
        # runtime settings + device + RCCL process group (dist.init_data_parallel); the model shards batches and evaluation
        dist.init_data_parallel(os.environ.get("HPMN_DP_BACKEND", "nccl"))
    if dataset_name == "amazon":
        trainset, testset, feature_size = load_dataset_pkl("../data/amazon/dataset_hpmn.pkl")
        model = Hpmn("model/amazon/hpmn/", trainset, testset, feature_size, 3, 2, 100, 100, 0.003, 32, 16, 3,
                     [2, 2, 5, 5, 1], [2, 2, 5, 5, 1], 3, 3, True, False, l2_reg=0., memory_reg=1e-5)
        model.train(2, 128)
        model.save_model()
    elif dataset_name == "taobao":
        trainset, testset, feature_size = load_dataset_pkl("../data/taobao/dataset_hpmn.pkl")
        model = Hpmn("model/taobao/hpmn/", trainset, testset, feature_size, 4, 3, 300, 36, 0.001, 32, 16, 3,
                     [2, 2, 3, 5, 5, 1], [2, 2, 3, 3, 1], 4, 5, True, False, l2_reg=0, memory_reg=1e-5)
        model.train(2, 128)
        model.save_model()
    elif dataset_name == "xlong":
        train_set = "../data/xlong/train_corpus_total_dual.txt"
        test_set = "../data/xlong/test_corpus_total_dual.txt"
        pv_cnt = 19002
        graph_emb = np.load("../data/xlong/graph_emb.npy")
        feature_size = pv_cnt + graph_emb.shape[0] + 20000
        emb_initializer = np.concatenate((graph_emb, np.zeros([20000, 16]), np.zeros([pv_cnt, 16])),
                                         0).astype(np.float32)
        model = Hpmn_Industry("model/xlong/hpmn/", train_set, test_set, feature_size, 2, 1, 1000 + 1, 184,
                              0.001, 32, 16, 3, [2] * 10 + [1], [3, 2, 2, 2, 2, 2, 2, 1], 5, 8, True, False,
                              emb_initializer, l2_reg=0, memory_reg=5e-5)
        model.train(epochs=3, batchsize=500)
        model.get_weights()
    else:
        print("Dataset must be one of taobao or amazon.")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
