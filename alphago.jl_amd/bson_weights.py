"""BSON weight checkpoints in the reference's on-disk format.

The reference saves a network as three BSON.jl documents of Flux parameter lists
(`save_model`, src/train.jl:14-35):

    models/weights/agz_base.bson    {"bn_weights":  [params(base_net)...]}
    models/weights/agz_value.bson   {"val_weights": [params(value)...]}
    models/weights/agz_policy.bson  {"pol_weights": [params(policy)...]}

and loads them back with `Flux.loadparams!` (`load_model`, src/play.jl:3-21).  Every entry is a
BSON.jl *tagged array*

    {tag: "array", type: {tag: "datatype", params: [], name: ["Core", "Float32"]},
     size: [d1, d2, ...], data: <binary, column-major>}

in Flux `params` order: Conv -> W [kw, kh, cin, cout], b;  BatchNorm -> beta, gamma;
Dense -> W [out, in], b.  Hence
    base   = stem conv (W, b), stem BN (beta, gamma), then per residual block
             conv1 (W, b), BN1 (beta, gamma), conv2 (W, b), BN2 (beta, gamma)   -> 4 + 8*tower arrays
    value  = conv 1x1 (W, b), BN (beta, gamma), Dense(N^2, 256) (W, b), Dense(256, 1) (W, b)
    policy = conv 1x1 (W, b), BN (beta, gamma), Dense(2N^2, A) (W, b)
BatchNorm running statistics and epsilon are not Flux params; they live in the struct dumps
models/agz_{base,value,policy}.bson (fields mu, sigma^2, eps of every Flux.BatchNorm struct) and are
read from there when present.

This module is host-side format code (the reference does the same on the host in Julia): a
minimal BSON reader/writer, no third-party dependency.  Arrays cross into the engine through
`agz_net_set_weights` in exactly the layout Flux stores them.
"""
import os
import struct

import numpy as np

from . import _lib

_F32 = {"tag": "datatype", "params": [], "name": ["Core", "Float32"]}


# ------------------------------------------------------------------ BSON (the subset BSON.jl emits)

class Binary(bytes):
    """BSON binary, subtype 0"""


def _parse_doc(b, off, as_list=False):
    size = struct.unpack_from("<i", b, off)[0]
    end = off + size
    if size < 5 or end > len(b) or b[end - 1] != 0:
        raise ValueError("malformed BSON document")
    off += 4
    keys, vals = [], []
    while b[off] != 0:
        t = b[off]
        z = b.index(0, off + 1)
        name = b[off + 1:z].decode("utf-8")
        off = z + 1
        if t == 0x01:
            v = struct.unpack_from("<d", b, off)[0]
            off += 8
        elif t == 0x02:
            n = struct.unpack_from("<i", b, off)[0]
            v = b[off + 4:off + 4 + n - 1].decode("utf-8")
            off += 4 + n
        elif t == 0x03:
            v, off = _parse_doc(b, off)
        elif t == 0x04:
            v, off = _parse_doc(b, off, as_list=True)
        elif t == 0x05:
            n = struct.unpack_from("<i", b, off)[0]
            v = Binary(b[off + 5:off + 5 + n])
            off += 5 + n
        elif t == 0x08:
            v = bool(b[off])
            off += 1
        elif t == 0x0A:
            v = None
        elif t == 0x10:
            v = struct.unpack_from("<i", b, off)[0]
            off += 4
        elif t == 0x12:
            v = struct.unpack_from("<q", b, off)[0]
            off += 8
        else:
            raise ValueError(f"unsupported BSON element type 0x{t:02x}")
        keys.append(name)
        vals.append(v)
    if off + 1 != end:
        raise ValueError("malformed BSON document (length mismatch)")
    return (vals if as_list else dict(zip(keys, vals))), end


def loads(data):
    doc, end = _parse_doc(bytes(data), 0)
    return doc


def _emit(v):
    """-> (type byte, payload)"""
    if isinstance(v, bool):
        return 0x08, b"\x01" if v else b"\x00"
    if isinstance(v, Binary):
        return 0x05, struct.pack("<i", len(v)) + b"\x00" + bytes(v)
    if isinstance(v, float):
        return 0x01, struct.pack("<d", v)
    if isinstance(v, (int, np.integer)):
        return 0x12, struct.pack("<q", int(v))      # BSON.jl writes Int64
    if isinstance(v, str):
        s = v.encode("utf-8") + b"\x00"
        return 0x02, struct.pack("<i", len(s)) + s
    if v is None:
        return 0x0A, b""
    if isinstance(v, dict):
        return 0x03, _emit_doc(v.items())
    if isinstance(v, (list, tuple)):
        return 0x04, _emit_doc((str(i), x) for i, x in enumerate(v))
    raise TypeError(type(v))


def _emit_doc(items):
    body = b""
    for k, v in items:
        t, payload = _emit(v)
        body += bytes([t]) + k.encode("utf-8") + b"\x00" + payload
    return struct.pack("<i", len(body) + 5) + body + b"\x00"


def dumps(doc):
    return _emit_doc(doc.items())


# ------------------------------------------------------------------ BSON.jl tagged arrays

def _is_f32_array(x):
    return isinstance(x, dict) and x.get("tag") == "array" and isinstance(x.get("type"), dict) \
        and x["type"].get("name") == ["Core", "Float32"]


def decode_array(x):
    """tagged array -> numpy float32 with the Julia shape (Fortran order)"""
    if not _is_f32_array(x):
        raise ValueError("not a BSON.jl Float32 array")
    shape = tuple(int(s) for s in x["size"])
    a = np.frombuffer(x["data"], dtype="<f4")
    if a.size != int(np.prod(shape, dtype=np.int64)):
        raise ValueError(f"array data length {a.size} does not match size {shape}")
    return a.reshape(shape, order="F").copy(order="F")


def encode_array(a):
    a = np.asarray(a, np.float32)
    return {"tag": "array", "type": _F32, "size": [int(s) for s in a.shape],
            "data": Binary(np.asfortranarray(a).tobytes(order="F"))}


def read_param_list(path, key=None):
    """one weights file -> list of numpy arrays in Flux `params` order"""
    doc = loads(open(path, "rb").read())
    if key is None:
        if len(doc) != 1:
            raise ValueError(f"{path}: expected one top-level key, found {list(doc)}")
        key = next(iter(doc))
    return [decode_array(x) for x in doc[key]]


def write_param_list(path, key, arrays):
    with open(path, "wb") as f:
        f.write(dumps({key: [encode_array(a) for a in arrays]}))


def _is_tracked(x):
    t = x.get("type") if isinstance(x, dict) else None
    return isinstance(t, dict) and t.get("name") == ["Flux", "Tracker", "TrackedArray"]


def batchnorm_stat_field(d):
    """which statistic the 5th field of a dumped Flux.BatchNorm holds.

    Flux <= 0.7 (Tracker era): struct (lambda, beta, gamma, mu, sigma, eps, momentum, active) with
    eps::Float64 = 1e-8 and TrackedArray parameters; test mode computes (x - mu) ./ sigma -- the field is
    the moving STANDARD DEVIATION with eps already folded in.  Flux >= 0.8: (..., mu, sigma2, eps::Float32 =
    1f-5, momentum, ...) and the forward divides by sqrt(sigma2 + eps) -- the field is the VARIANCE.
    The files shipped with the reference (models/agz_*.bson) are of the first kind."""
    tracked = _is_tracked(d[1]) or _is_tracked(d[2])
    eps = d[5]
    old_eps = isinstance(eps, float) and abs(eps - 1e-8) < 1e-12
    return "std" if (tracked or old_eps) else "var"


def read_batchnorm_stats(path, bn_field="auto"):
    """walk a `@save`d model struct dump and return [(mu, sigma2, eps), ...] for every Flux.BatchNorm in
    depth-first (= layer) order, ALWAYS as (mean, variance, eps) such that the inference affine is
    gamma / sqrt(sigma2 + eps).  bn_field: "auto" (detect the Flux generation per layer, see
    batchnorm_stat_field), "std" (field 5 is sigma: returns sigma^2 and eps = 0, i.e. exactly
    (x - mu) / sigma) or "var" (field 5 is sigma^2, eps kept)."""
    if bn_field not in ("auto", "std", "var"):
        raise ValueError('bn_field must be "auto", "std" or "var"')
    doc = loads(open(path, "rb").read())
    out = []

    def walk(x):
        if isinstance(x, dict):
            t = x.get("type")
            if x.get("tag") == "struct" and isinstance(t, dict) and t.get("name") == ["Flux", "BatchNorm"]:
                d = x["data"]
                mu, field, eps = decode_array(d[3]).ravel(), decode_array(d[4]).ravel(), float(d[5])
                kind = batchnorm_stat_field(d) if bn_field == "auto" else bn_field
                if kind == "std":
                    out.append((mu, (field.astype(np.float64) ** 2).astype(np.float32), 0.0))
                else:
                    out.append((mu, field, eps))
                return
            for v in x.values():
                walk(v)
        elif isinstance(x, list):
            for v in x:
                walk(v)

    walk(doc)
    return out


# ------------------------------------------------------------------ mapping onto the engine's layers

def _check(a, shape, what):
    if tuple(a.shape) != tuple(shape):
        raise ValueError(f"{what}: shape {tuple(a.shape)}, expected {tuple(shape)}")


def tower_height_of(base_list):
    n = len(base_list)
    if n < 4 or (n - 4) % 8:
        raise ValueError(f"base parameter list has {n} arrays; expected 4 + 8*tower_height")
    return (n - 4) // 8


def apply_param_lists(engine, base, value, policy, base_stats=None, value_stats=None, policy_stats=None):
    """push Flux-ordered parameter lists into an Engine (the role of Flux.loadparams!, play.jl:13-15)"""
    N = engine.N
    P, A = N * N, N * N + 1
    t = tower_height_of(base)
    if t != engine.tower_height:
        raise ValueError(f"checkpoint has tower_height {t}, engine was built with {engine.tower_height}")
    K = _lib

    def conv_bn(layer, arrs, cin, cout, k, stats, what):
        W, b, beta, gamma = arrs
        _check(W, (k, k, cin, cout), what + " W")
        for a, nm in ((b, "b"), (beta, "beta"), (gamma, "gamma")):
            _check(a.reshape(-1), (cout,), f"{what} {nm}")
        engine.set_weights(layer, K.K_WEIGHT, W.ravel(order="F"))
        engine.set_weights(layer, K.K_BIAS, b.ravel())
        engine.set_weights(layer, K.K_BN_BETA, beta.ravel())
        engine.set_weights(layer, K.K_BN_GAMMA, gamma.ravel())
        if stats is not None:
            mu, var, eps = stats
            _check(mu, (cout,), what + " mu")
            _check(var, (cout,), what + " sigma2")
            engine.set_weights(layer, K.K_BN_MEAN, mu)
            engine.set_weights(layer, K.K_BN_VAR, var)
            engine.set_weights(layer, K.K_BN_EPS, np.array([eps], np.float32))

    def dense(layer, W, b, out, inn, what):
        _check(W, (out, inn), what + " W")
        _check(b.reshape(-1), (out,), what + " b")
        engine.set_weights(layer, K.K_WEIGHT, W.ravel(order="F"))
        engine.set_weights(layer, K.K_BIAS, b.ravel())

    st = (lambda s, i: None if s is None else s[i])
    if base_stats is not None and len(base_stats) != 1 + 2 * t:
        raise ValueError("base BatchNorm statistics do not match the tower height")
    conv_bn(0, base[0:4], 17, 256, 3, st(base_stats, 0), "stem")
    for l in range(2 * t):
        conv_bn(1 + l, base[4 + 4 * l:8 + 4 * l], 256, 256, 3, st(base_stats, 1 + l), f"tower conv {l}")
    if len(value) != 8 or len(policy) != 6:
        raise ValueError("value/policy parameter lists must have 8/6 arrays")
    conv_bn(K.L_VALUE_CONV, value[0:4], 256, 1, 1, st(value_stats, 0), "value conv")
    dense(K.L_VALUE_FC1, value[4], value[5], 256, P, "value Dense 1")
    dense(K.L_VALUE_FC2, value[6], value[7], 1, 256, "value Dense 2")
    conv_bn(K.L_POLICY_CONV, policy[0:4], 256, 2, 1, st(policy_stats, 0), "policy conv")
    dense(K.L_POLICY_FC, policy[4], policy[5], A, 2 * P, "policy Dense")


def extract_param_lists(engine):
    """the inverse: Flux-ordered parameter lists (and BN statistics) read back from the engine"""
    N, t = engine.N, engine.tower_height
    P, A = N * N, N * N + 1
    K = _lib

    def conv_bn(layer, cin, cout, k):
        W = engine.get_weights(layer, K.K_WEIGHT).reshape((k, k, cin, cout), order="F")
        arrs = [W] + [engine.get_weights(layer, kk) for kk in (K.K_BIAS, K.K_BN_BETA, K.K_BN_GAMMA)]
        stats = (engine.get_weights(layer, K.K_BN_MEAN), engine.get_weights(layer, K.K_BN_VAR),
                 float(engine.get_weights(layer, K.K_BN_EPS)[0]))
        return arrs, stats

    def dense(layer, out, inn):
        return [engine.get_weights(layer, K.K_WEIGHT).reshape((out, inn), order="F"), engine.get_weights(layer, K.K_BIAS)]

    base, base_stats = [], []
    for l in range(1 + 2 * t):
        a, s = conv_bn(l, 17 if l == 0 else 256, 256, 3)
        base += a
        base_stats.append(s)
    va, vs = conv_bn(K.L_VALUE_CONV, 256, 1, 1)
    value = va + dense(K.L_VALUE_FC1, 256, P) + dense(K.L_VALUE_FC2, 1, 256)
    pa, ps = conv_bn(K.L_POLICY_CONV, 256, 2, 1)
    policy = pa + dense(K.L_POLICY_FC, A, 2 * P)
    return dict(base=base, value=value, policy=policy, base_stats=base_stats, value_stats=[vs], policy_stats=[ps])


def read_checkpoint(model_dir, bn_field="auto"):
    """`load_model(str, env)` (src/play.jl:3-21): str/weights/agz_*.bson + BatchNorm statistics.  The
    statistics come from the side files weights/agz_*_bnstats.bson when write_checkpoint has put them
    there (they belong to exactly these weights), else from the reference-style struct dumps
    str/agz_*.bson (bn_field: how their 5th BatchNorm field is read, see read_batchnorm_stats)."""
    w = os.path.join(model_dir, "weights")
    out = dict(base=read_param_list(os.path.join(w, "agz_base.bson")),
               value=read_param_list(os.path.join(w, "agz_value.bson")),
               policy=read_param_list(os.path.join(w, "agz_policy.bson")))
    for part in ("base", "value", "policy"):
        side = os.path.join(w, f"agz_{part}_bnstats.bson")
        dump = os.path.join(model_dir, f"agz_{part}.bson")
        if os.path.exists(side):
            out[part + "_stats"] = [
                (decode_array(d["mu"]).ravel(), decode_array(d["sigma2"]).ravel(), float(d["eps"]))
                for d in loads(open(side, "rb").read())["stats"]]
        elif os.path.exists(dump):
            out[part + "_stats"] = read_batchnorm_stats(dump, bn_field)
        else:
            out[part + "_stats"] = None
    return out


def write_checkpoint(model_dir, lists):
    """`save_model(nn)` (src/train.jl:14-35), weights part: the three parameter-list files with the
    reference's keys.  The struct dump `agz_model.bson` is a serialised Flux object graph and is not
    reproduced; BatchNorm statistics go into side files agz_*_bnstats.bson instead."""
    w = os.path.join(model_dir, "weights")
    os.makedirs(w, exist_ok=True)
    write_param_list(os.path.join(w, "agz_base.bson"), "bn_weights", lists["base"])
    write_param_list(os.path.join(w, "agz_value.bson"), "val_weights", lists["value"])
    write_param_list(os.path.join(w, "agz_policy.bson"), "pol_weights", lists["policy"])
    for part in ("base", "value", "policy"):
        stats = lists.get(part + "_stats")
        if stats is None:
            continue
        doc = {"stats": [{"mu": encode_array(m), "sigma2": encode_array(v), "eps": float(e)} for m, v, e in stats]}
        with open(os.path.join(w, f"agz_{part}_bnstats.bson"), "wb") as f:
            f.write(dumps(doc))
