"""TensorFlow-1 checkpoint interop: read and write the tensor-bundle files ``tf.train.Saver`` produces
(``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` + the ``checkpoint`` state file), with the variable names the
reference graph would have under TF 1.4 -- ``saver.save`` / ``saver.restore`` of code/hpmn.py:61, :91-92, :105-111.

Why it exists: the arithmetic of the reference lives in TF 1.4, which cannot run in this environment, so parity is
pinned only by restating it.  With this module a TF 1.4 run of code/hpmn.py elsewhere can hand its
trained weights to this implementation (``Hpmn.load_model``), or take ours (``Hpmn.save_model``), which is the one
route to a true cross-implementation check.

Formats (restated from the published TensorFlow / LevelDB sources; nothing of either is vendored):
  * data file: the tensors' raw little-endian bytes, back to back, in key order;
  * index file: a LevelDB-format table (tensorflow/core/lib/io/table*, identical to leveldb/table/format.h):
    blocks of prefix-compressed (key, value) entries + restart array, each followed by a 5-byte trailer (compression
    type 0, masked crc32c), a meta-index block, an index block, a 48-byte footer ending in the magic
    0xdb4775248b80fb57.  Key "" holds a BundleHeaderProto, every other key a BundleEntryProto (dtype, shape, shard,
    offset, size, masked crc32c of the bytes) -- tensorflow/core/protobuf/tensor_bundle.proto.
The reader handles any such table (multiple blocks, prefix compression, any restart interval, snappy excluded); the
writer emits the plainest valid one (4 KiB blocks, restart interval 16, like TF's defaults).

Variable names.  TF 1.4 scopes as written at code/hpmn.py:117 (``GRU%s``), :119 (dynamic_rnn -> ``rnn``, GRUCell ->
``gru_cell/{gates,candidate}/{kernel,bias}``), :173-174 (``dense``, ``map``), :137-139 (``dense_1`` ...), :190-195
(``bn1``, ``fc1..3`` under ``output``), :433-464 / :285-320 (``Embedding``, ``User``, ``item`` / ``Item``).  Adam's
slots (``<var>/Adam``, ``<var>/Adam_1``) and ``beta{1,2}_power`` are created inside the ``output`` variable scope
(:209-214 runs under :464), hence their ``output/`` prefix.  These names are inferred from the scopes, not verified
against a TF run (there is none to be had here); ``read_bundle`` returns whatever the file holds, and
``tf_name_map`` is the single place to adjust.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
from typing import Dict, Iterable, List, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
DT_FLOAT, DT_INT32, DT_INT64, DT_DOUBLE = 1, 3, 9, 2
_DTYPES = {DT_FLOAT: np.dtype("<f4"), DT_INT32: np.dtype("<i4"), DT_INT64: np.dtype("<i8"), DT_DOUBLE: np.dtype("<f8")}
_DT_OF = {np.dtype("float32"): DT_FLOAT, np.dtype("int32"): DT_INT32, np.dtype("int64"): DT_INT64,
          np.dtype("float64"): DT_DOUBLE}


# ------------------------------------------------------------------------------------------------ crc32c
_host = None


def _host_lib():
    global _host
    if _host is None:
        try:
            from . import build
            lib = C.CDLL(build.build_host_library())
            lib.hpmn_crc32c_extend.restype = C.c_uint32
            lib.hpmn_crc32c_extend.argtypes = [C.c_uint32, C.c_void_p, C.c_size_t]
            _host = lib
        except Exception:                      # no C compiler: the (slow) table loop below
            _host = False
    return _host


_TAB = None


def crc32c(data, crc: int = 0) -> int:
    """CRC-32C of ``data`` (bytes-like or C-contiguous ndarray), continuing from ``crc``."""
    arr = np.frombuffer(memoryview(data).cast("B"), dtype=np.uint8) if not isinstance(data, np.ndarray) \
        else np.ascontiguousarray(data).reshape(-1).view(np.uint8)
    lib = _host_lib()
    if lib:
        return int(lib.hpmn_crc32c_extend(crc, arr.ctypes.data, arr.size)) if arr.size else crc
    global _TAB
    if _TAB is None:
        _TAB = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            _TAB.append(c)
    c = crc ^ 0xFFFFFFFF
    for b in arr.tobytes():
        c = (c >> 8) ^ _TAB[(c ^ b) & 0xFF]
    return c ^ 0xFFFFFFFF


def mask_crc(crc: int) -> int:
    """leveldb/TF crc masking: rotate right by 15 and add a constant (crc32c.h Mask)."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ varints / protobuf
def _put_varint(x: int) -> bytes:
    out = bytearray()
    x &= (1 << 64) - 1
    while x >= 0x80:
        out.append((x & 0x7F) | 0x80)
        x >>= 7
    out.append(x)
    return bytes(out)


def _get_varint(buf, pos: int) -> Tuple[int, int]:
    x, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        x |= (b & 0x7F) << shift
        if b < 0x80:
            return x, pos
        shift += 7


def _pb_fields(buf) -> Iterable[Tuple[int, int, object]]:
    """(field number, wire type, value) of a serialized protobuf message (wire types 0, 1, 2, 5)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _encode_entry(dtype: int, shape: Tuple[int, ...], offset: int, size: int, crc_masked: int) -> bytes:
    """BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32)}."""
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(s) for s in shape))
    out = b"\x08" + _put_varint(dtype) + b"\x12" + _put_varint(len(dims)) + dims
    if offset:
        out += b"\x20" + _put_varint(offset)
    out += b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc_masked)
    return out


def _decode_entry(buf) -> dict:
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=0, sliced=False)
    for f, _, v in _pb_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:
            dims = []
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3 if v3 < (1 << 63) else v3 - (1 << 64)
                    dims.append(size)
            e["shape"] = tuple(dims)
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc32c"] = v
        elif f == 7:
            e["sliced"] = True
    return e


HEADER_PROTO = b"\x08\x01" + b"\x1a\x02\x08\x01"      # BundleHeaderProto{num_shards: 1, (endianness: LITTLE = 0), version{producer: 1}}


# ------------------------------------------------------------------------------------------------ table (leveldb format)
def _build_block(entries: List[Tuple[bytes, bytes]], restart_interval: int = 16) -> bytes:
    out, restarts, last = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            m = min(len(k), len(last))
            while shared < m and k[shared] == last[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _block_with_trailer(block: bytes) -> bytes:
    crc = crc32c(b"\x00", crc32c(block))        # crc over contents + the compression-type byte (0 = none)
    return block + b"\x00" + struct.pack("<I", mask_crc(crc))


def _parse_block(buf) -> List[Tuple[bytes, bytes]]:
    n_restarts = struct.unpack_from("<I", buf, len(buf) - 4)[0]
    end = len(buf) - 4 - 4 * n_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _get_varint(buf, pos)
        unshared, pos = _get_varint(buf, pos)
        vlen, pos = _get_varint(buf, pos)
        key = key[:shared] + bytes(buf[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(buf[pos:pos + vlen])))
        pos += vlen
    return out


def _read_block(data, offset: int, size: int, verify: bool = True) -> bytes:
    block = data[offset:offset + size]
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("compressed table block (type %d): snappy is not supported" % ctype)
    if verify:
        want = struct.unpack_from("<I", data, offset + size + 1)[0]
        if mask_crc(crc32c(bytes([ctype]), crc32c(block))) != want:
            raise ValueError("table block checksum mismatch at offset %d" % offset)
    return block


def write_table(path: str, items: List[Tuple[bytes, bytes]], block_size: int = 4096) -> None:
    """A LevelDB-format table holding ``items`` (sorted by key)."""
    assert all(items[i][0] < items[i + 1][0] for i in range(len(items) - 1)), "keys must be sorted and unique"
    out = bytearray()
    index: List[Tuple[bytes, bytes]] = []
    cur: List[Tuple[bytes, bytes]] = []
    cur_bytes = 0

    def flush():
        nonlocal cur, cur_bytes
        if not cur:
            return
        block = _build_block(cur)
        handle = _put_varint(len(out)) + _put_varint(len(block))
        index.append((cur[-1][0], handle))           # separator: the block's last key (>= every key in it)
        out.extend(_block_with_trailer(block))
        cur, cur_bytes = [], 0

    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 3
        if cur_bytes >= block_size:
            flush()
    flush()
    meta = _build_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out.extend(_block_with_trailer(meta))
    idx = _build_block(index, restart_interval=1)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out.extend(_block_with_trailer(idx))
    footer = meta_handle + idx_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(bytes(out))


def read_table(path: str) -> List[Tuple[bytes, bytes]]:
    data = open(path, "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s is not a TensorFlow checkpoint index (bad table magic)" % path)
    foot = data[len(data) - 48:]
    pos = 0
    _, pos = _get_varint(foot, pos)       # metaindex handle
    _, pos = _get_varint(foot, pos)
    i_off, pos = _get_varint(foot, pos)
    i_size, pos = _get_varint(foot, pos)
    out = []
    for _, handle in _parse_block(_read_block(data, i_off, i_size)):
        off, p2 = _get_varint(handle, 0)
        size, _ = _get_varint(handle, p2)
        out.extend(_parse_block(_read_block(data, off, size)))
    return out


# ------------------------------------------------------------------------------------------------ bundles
def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
    """``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` for the named tensors (one shard), plus the
    ``checkpoint`` state file tf.train.latest_checkpoint reads."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    items: List[Tuple[bytes, bytes]] = [(b"", HEADER_PROTO)]
    offset = 0
    tmp = prefix + ".data-00000-of-00001.tmp"
    with open(tmp, "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.require(tensors[name], requirements="C")          # (keeps 0-d scalars 0-d)
            if np.dtype(a.dtype.name) not in _DT_OF:
                raise TypeError("%s: dtype %s is not supported" % (name, a.dtype))
            a = a.astype(a.dtype.newbyteorder("<"), copy=False)
            flat = a.reshape(-1)
            f.write(memoryview(flat).cast("B"))
            items.append((name.encode(), _encode_entry(_DT_OF[np.dtype(a.dtype.name)], tuple(a.shape), offset, a.nbytes,
                                                       mask_crc(crc32c(flat)))))
            offset += a.nbytes
    os.replace(tmp, prefix + ".data-00000-of-00001")
    write_table(prefix + ".index", items)
    base = os.path.basename(prefix)
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def read_bundle(prefix: str, names: Optional[Iterable[str]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    """All (or the named) tensors of a TF tensor bundle, checksums verified."""
    entries = read_table(prefix + ".index")
    if not entries or entries[0][0] != b"":
        raise ValueError("%s.index has no bundle header" % prefix)
    num_shards = 1
    for f, _, v in _pb_fields(entries[0][1]):
        if f == 1:
            num_shards = v
        if f == 2 and v != 0:
            raise ValueError("big-endian bundle")
    want = None if names is None else set(names)
    out: Dict[str, np.ndarray] = {}
    files: Dict[int, object] = {}
    try:
        for key, val in entries[1:]:
            name = key.decode()
            if want is not None and name not in want:
                continue
            e = _decode_entry(val)
            if e["sliced"]:
                raise ValueError("%s: partitioned (sliced) variables are not supported" % name)
            if e["dtype"] not in _DTYPES:
                raise ValueError("%s: dtype enum %d is not supported" % (name, e["dtype"]))
            if e["shard_id"] not in files:
                files[e["shard_id"]] = open("%s.data-%05d-of-%05d" % (prefix, e["shard_id"], num_shards), "rb")
            fh = files[e["shard_id"]]
            fh.seek(e["offset"])
            raw = fh.read(e["size"])
            if len(raw) != e["size"]:
                raise ValueError("%s: data file is truncated" % name)
            if verify and mask_crc(crc32c(raw)) != e["crc32c"]:
                raise ValueError("%s: tensor checksum mismatch" % name)
            out[name] = np.frombuffer(raw, dtype=_DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    finally:
        for fh in files.values():
            fh.close()
    return out


# ------------------------------------------------------------------------------------------------ names
def tf_name_map(param_names: Iterable[str]) -> Dict[str, str]:
    """our variable name -> the name TF 1.4 gives it (module docstring).  Only the GRU cells differ: the cell's
    variables sit under dynamic_rnn's ``rnn`` scope and the cell's own ``gru_cell``."""
    out = {}
    for n in param_names:
        parts = n.split("/")
        if len(parts) >= 4 and parts[1].startswith("GRU") and parts[2] in ("gates", "candidate"):
            out[n] = "/".join(parts[:2] + ["rnn", "gru_cell"] + parts[2:])
        else:
            out[n] = n
    return out


ADAM_SCOPE = "output"        # the optimizer is created inside variable_scope('output') (code/hpmn.py:464, :209-214)
STEP_TENSOR = "hpmn_amd/adam_step"      # int64 scalar, ours only: the exact number of Adam steps taken


def export_model(prefix: str, params: Dict[str, np.ndarray], adam_m: Optional[Dict[str, np.ndarray]] = None,
                 adam_v: Optional[Dict[str, np.ndarray]] = None, adam_t: int = 0, beta1: float = 0.9,
                 beta2: float = 0.999, mask_table_rows: Optional[int] = None) -> None:
    """What ``tf.train.Saver().save(sess, prefix)`` of the reference graph would hold for these variables: the
    trainable variables, batch-norm's (never updated) moving statistics, Hpmn's constant mask table, Adam's slots
    and the beta powers (after t steps TF holds beta^(t+1))."""
    names = tf_name_map(params)
    t: Dict[str, np.ndarray] = {}
    for k, v in params.items():
        t[names[k]] = np.asarray(v, dtype=np.float32)
        if k.endswith("bn1/gamma"):
            scope = names[k][:-len("gamma")]
            t[scope + "moving_mean"] = np.zeros(v.shape, np.float32)          # never updated: BN runs in inference
            t[scope + "moving_variance"] = np.ones(v.shape, np.float32)       # mode and no update op is ever fetched
    if mask_table_rows is not None:                                           # code/hpmn.py:417-419
        mt = np.ones((mask_table_rows, 1), np.float32)
        mt[0] = 0.0
        t["Embedding/mask_lookup_table"] = mt
    if adam_m is not None and adam_v is not None:
        for k in params:
            t["%s/%s/Adam" % (ADAM_SCOPE, names[k])] = np.asarray(adam_m[k], dtype=np.float32)
            t["%s/%s/Adam_1" % (ADAM_SCOPE, names[k])] = np.asarray(adam_v[k], dtype=np.float32)
        t[ADAM_SCOPE + "/beta1_power"] = np.asarray(beta1 ** (adam_t + 1), dtype=np.float32)
        t[ADAM_SCOPE + "/beta2_power"] = np.asarray(beta2 ** (adam_t + 1), dtype=np.float32)
        # float32 beta1^(t+1) underflows to 0 near t = 985 (TF's own variable does too, and its lr_t then is lr sqrt(1 -
        # b2^t)): the step count is ALSO stored exactly, under a name the TF graph does not have (Saver.restore ignores
        # tensors no variable asks for)
        t[STEP_TENSOR] = np.asarray(adam_t, dtype=np.int64)
    write_bundle(prefix, t)


def import_model(prefix: str, param_shapes: Dict[str, Tuple[int, ...]], beta1: float = 0.9, beta2: float = 0.999):
    """-> (params, adam_m or None, adam_v or None, adam_t).  Every variable of ``param_shapes`` must be in the
    bundle with that shape (extra tensors -- the never-executed item branch of a reference checkpoint, BN's moving
    statistics -- are ignored); Adam state is optional (a checkpoint of weights only restores with t = 0)."""
    have = read_bundle(prefix)
    names = tf_name_map(param_shapes)
    params, m, v = {}, {}, {}
    for k, shape in param_shapes.items():
        tn = names[k]
        if tn not in have:
            raise KeyError("checkpoint %s has no tensor %r (for %r)" % (prefix, tn, k))
        if tuple(have[tn].shape) != tuple(shape):
            raise ValueError("%s: checkpoint shape %s, model shape %s" % (tn, have[tn].shape, tuple(shape)))
        params[k] = have[tn]
        a, b = "%s/%s/Adam" % (ADAM_SCOPE, tn), "%s/%s/Adam_1" % (ADAM_SCOPE, tn)
        if a in have and b in have:
            m[k], v[k] = have[a], have[b]
    if len(m) != len(param_shapes):
        return params, None, None, 0
    return params, m, v, _adam_steps(have, beta1, beta2)


def _adam_steps(have, beta1: float, beta2: float) -> int:
    """Adam's step count of a checkpoint: the exact tensor our writer adds; for a checkpoint TF wrote, from beta2_power
    (float32 0.999^(t+1) resolves t to about 87 k steps; beta1_power underflows to 0 near t = 985), then beta1_power.
    A power that is 0 or below float32's smallest normal has underflowed: TF's lr_t = lr sqrt(1 - b2^t) / (1 - b1^t) is
    then lr itself, so the count returned SATURATES (log(tiny)/log(beta2), about 87 k for 0.999) instead of falling
    back to 0, which would restart the bias correction (lr_t = 0.32 lr) on late-step moments."""
    if STEP_TENSOR in have:
        return max(0, int(np.asarray(have[STEP_TENSOR]).reshape(-1)[0]))
    tiny = float(np.finfo(np.float32).tiny)
    underflowed = False
    for name, beta in ((ADAM_SCOPE + "/beta2_power", beta2), (ADAM_SCOPE + "/beta1_power", beta1)):
        p = have.get(name)
        if p is None:
            continue
        p = float(np.asarray(p).reshape(-1)[0])
        if np.isfinite(p) and tiny <= p < 1.0:
            return max(0, int(round(np.log(p) / np.log(beta))) - 1)
        if np.isfinite(p) and 0.0 <= p < tiny:
            underflowed = True
    if underflowed:
        return int(np.log(tiny) / np.log(beta2))
    return 0
