"""Weight checkpoints in the reference's BSON format (src/train.jl:14-35,
src/play.jl:3-21).  CPU: reader/writer round trips, shapes and ordering, byte-exact rewrite of the
files shipped with the reference (only where /root/reference exists), oracle forward with the
shipped 9x9/tower-0 parameters against the committed fixture.  GPU: load_model -> forward within
1e-4 of the float64 fixture; save_model writes back the same bytes."""
import os
import tempfile

import numpy as np
import pytest

import alphago_jl_amd as ag
import orc
from golden.make_golden import OracleNetSink

bw = ag.bson_weights
L = orc.lib()
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_MODELS = "/root/reference/models"


def fixture_lists():
    d = np.load(os.path.join(G, "shipped_9x9_t0.npz"))
    out = {}
    for part, n in (("base", 4), ("value", 8), ("policy", 6)):
        out[part] = [d[f"{part}_{i}"] for i in range(n)]
        out[part + "_stats"] = [(d[f"{part}_mu_0"], d[f"{part}_var_0"], float(d[f"{part}_eps_0"][0]))]
    return d, out


def test_bson_subset_roundtrip():
    doc = {"a": [1, 2.5, "x", None, True, {"b": bw.Binary(b"\x00\x01\x02")}], "n": -7}
    back = bw.loads(bw.dumps(doc))
    assert back == doc
    with pytest.raises(ValueError):
        bw.loads(bw.dumps(doc)[:-3])


def test_tagged_array_is_column_major():
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    enc = bw.encode_array(a)
    assert enc["size"] == [2, 3, 4] and enc["type"]["name"] == ["Core", "Float32"]
    # Julia linear index 2 (1-based) is element [2,1,1] -> a[1,0,0]
    assert np.frombuffer(enc["data"], "<f4")[1] == a[1, 0, 0]
    assert (bw.decode_array(enc) == a).all()
    with pytest.raises(ValueError):
        bw.decode_array({**enc, "size": [5, 5]})


def test_param_list_files_and_tower_height():
    rng = np.random.RandomState(0)
    t = 2
    base = [rng.randn(3, 3, 17, 256), rng.randn(256), rng.randn(256), rng.randn(256)]
    for _ in range(2 * t):
        base += [rng.randn(3, 3, 256, 256), rng.randn(256), rng.randn(256), rng.randn(256)]
    base = [a.astype(np.float32) for a in base]
    assert bw.tower_height_of(base) == t
    with pytest.raises(ValueError):
        bw.tower_height_of(base[:-1])
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "agz_base.bson")
        bw.write_param_list(p, "bn_weights", base)
        back = bw.read_param_list(p)
        assert len(back) == len(base) and all((x == y).all() and x.shape == y.shape for x, y in zip(back, base))
        assert list(bw.loads(open(p, "rb").read())) == ["bn_weights"]          # the key train.jl:33 uses


def test_checkpoint_roundtrip_with_bn_stats():
    _, lists = fixture_lists()
    with tempfile.TemporaryDirectory() as d:
        bw.write_checkpoint(d, lists)
        ck = bw.read_checkpoint(d)
        for part in ("base", "value", "policy"):
            assert all((x == y).all() for x, y in zip(ck[part], lists[part]))
            (m, v, e), (m2, v2, e2) = ck[part + "_stats"][0], lists[part + "_stats"][0]
            assert (m == m2).all() and (v == v2).all() and e == e2 == 0.0
        # save -> load is a round trip even when stale reference-style struct dumps sit in the directory:
        # the side files written with the weights win
        for part in ("base", "value", "policy"):
            with open(os.path.join(d, f"agz_{part}.bson"), "wb") as f:
                f.write(bw.dumps({"x": _bn_dump(np.ones(1), np.full(1, 9.0), 1e-8, old=True)}))
        ck = bw.read_checkpoint(d)
        assert all((ck[p + "_stats"][0][1] == lists[p + "_stats"][0][1]).all() for p in ("base", "value", "policy"))


def _bn_dump(mu, field, eps, old):
    """a dumped Flux.BatchNorm struct in the two generations (field order of both: lambda, beta, gamma, mu,
    sigma | sigma2, eps, momentum, active)"""
    arr = lambda a: bw.encode_array(np.asarray(a, np.float32))
    par = (lambda a: {"tag": "struct", "type": {"tag": "datatype", "name": ["Flux", "Tracker", "TrackedArray"], "params": []},
                      "data": [arr(a)]}) if old else arr
    return {"tag": "struct", "type": {"tag": "datatype", "name": ["Flux", "BatchNorm"], "params": []},
            "data": [{"tag": "struct", "type": {"tag": "datatype", "name": ["NNlib", "#relu"], "params": []}, "data": []},
                     par(np.zeros_like(mu)), par(np.ones_like(mu)), arr(mu), arr(field), eps, 0.1, True]}


def test_batchnorm_field_generation_is_detected_and_switchable():
    """Flux <= 0.7 dumps (TrackedArray parameters, Float64 eps = 1e-8) carry sigma; Flux >= 0.8 carry sigma^2.
    read_batchnorm_stats always returns (mean, variance, eps) for gamma / sqrt(variance + eps)."""
    mu, field = np.array([0.5, -1.0], np.float32), np.array([0.25, 2.0], np.float32)
    with tempfile.TemporaryDirectory() as d:
        po, pn = os.path.join(d, "old.bson"), os.path.join(d, "new.bson")
        open(po, "wb").write(bw.dumps({"bn": {"layers": [_bn_dump(mu, field, 1e-8, old=True)]}}))
        open(pn, "wb").write(bw.dumps({"bn": {"layers": [_bn_dump(mu, field, float(np.float32(1e-5)), old=False)]}}))
        (m, v, e), = bw.read_batchnorm_stats(po)                    # auto -> std
        assert (m == mu).all() and np.allclose(v, field ** 2, rtol=1e-7) and e == 0.0
        (m, v, e), = bw.read_batchnorm_stats(pn)                    # auto -> var
        assert (v == field).all() and e == pytest.approx(1e-5)
        (m, v, e), = bw.read_batchnorm_stats(po, "var")             # explicit override, both ways
        assert (v == field).all() and e == 1e-8
        (m, v, e), = bw.read_batchnorm_stats(pn, "std")
        assert np.allclose(v, field ** 2, rtol=1e-7) and e == 0.0
        with pytest.raises(ValueError):
            bw.read_batchnorm_stats(po, "sigma")


def test_shipped_stem_activations_have_unit_scale_under_the_std_reading():
    """the check that tells sigma from sigma^2 without Flux: a trained BatchNorm normalises its input, so on
    real positions the stem's pre-affine activations (conv(x) + b - mu) / sigma must have per-channel spread of
    order 1.  Read as a variance, the same field gives spreads of sqrt(sigma) ~ 0.2-0.9."""
    import torch
    from test_hostsim_go import random_positions
    d, lists = fixture_lists()
    W, b = lists["base"][0], lists["base"][1]                     # [3,3,17,256] Flux layout, true convolution
    positions = random_positions(9, 24, 60, seed=5)
    x = np.stack([orc.feats(p).reshape(17, 9, 9) for p in positions]).astype(np.float64)    # [B,17,col,row] col-major
    w = torch.from_numpy(np.ascontiguousarray(W[::-1, ::-1].transpose(3, 2, 1, 0)).astype(np.float64))
    y = torch.nn.functional.conv2d(torch.from_numpy(x), w, padding=1).numpy() + b.reshape(1, -1, 1, 1)
    mu, sigma = d["base_mu_0"].astype(np.float64), d["base_field_0"].astype(np.float64)
    spread = y.transpose(1, 0, 2, 3).reshape(256, -1).std(1)
    live = spread > 1e-3
    ratio_std = np.median(spread[live] / sigma[live])
    ratio_var = np.median(spread[live] / np.sqrt(sigma[live] + 1e-8))
    print(f"median activation spread / sigma = {ratio_std:.3f}; / sqrt(field) = {ratio_var:.3f}")
    assert 0.6 < ratio_std < 1.4
    assert abs(np.log(ratio_std)) < abs(np.log(ratio_var))
    centred = np.abs(y.transpose(1, 0, 2, 3).reshape(256, -1).mean(1) - mu)[live] / sigma[live]
    assert np.median(centred) < 0.5


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="reference checkout not present")
def test_shipped_files_decode_and_rewrite_byte_exact():
    ck = bw.read_checkpoint(REF_MODELS)
    d, lists = fixture_lists()
    for part in ("base", "value", "policy"):
        assert all((x == y).all() and x.shape == y.shape for x, y in zip(ck[part], lists[part]))
        m, v, e = ck[part + "_stats"][0]                       # Flux <= 0.7 dump: field 5 is sigma -> (sigma^2, 0)
        assert (m == lists[part + "_stats"][0][0]).all() and (v == lists[part + "_stats"][0][1]).all() and e == 0.0
        raw = bw.read_batchnorm_stats(os.path.join(REF_MODELS, f"agz_{part}.bson"), "var")[0]
        assert (raw[1] == d[f"{part}_field_0"]).all() and raw[2] == 1e-8
        assert np.allclose(v, raw[1].astype(np.float64) ** 2, rtol=1e-6)
    with tempfile.TemporaryDirectory() as t:
        bw.write_checkpoint(t, ck)
        for part in ("base", "value", "policy"):
            a = open(os.path.join(REF_MODELS, "weights", f"agz_{part}.bson"), "rb").read()
            b = open(os.path.join(t, "weights", f"agz_{part}.bson"), "rb").read()
            assert a == b, part


def test_oracle_forward_with_shipped_weights_matches_fixture():
    d, lists = fixture_lists()
    sink = OracleNetSink(9, 0)
    bw.apply_param_lists(sink, lists["base"], lists["value"], lists["policy"], lists["base_stats"],
                         lists["value_stats"], lists["policy_stats"])
    x = d["feats"].astype(np.float32)
    B = x.shape[0]
    pi, v = np.zeros((B, 82), np.float32), np.zeros(B, np.float32)
    L.or_net_forward_feats(sink.net, orc.fptr(x), B, orc.fptr(pi), orc.fptr(v), 64)
    assert np.abs(pi - d["pi_f64"]).max() < 1e-12 and np.abs(v - d["v_f64"]).max() < 1e-12
    assert np.allclose(pi.sum(1), 1, atol=1e-5)
    # shape errors are reported, not swallowed
    bad = [a.copy() for a in lists["value"]]
    bad[4] = bad[4][:, :80]
    with pytest.raises(ValueError):
        bw.apply_param_lists(sink, lists["base"], bad, lists["policy"])
    L.or_net_free(sink.net)


@pytest.mark.gpu
def test_gpu_load_model_forward_and_save_model():
    d, lists = fixture_lists()
    env = ag.GoEnv(9)
    with tempfile.TemporaryDirectory() as t:
        bw.write_checkpoint(t, lists)
        nn = ag.load_model(t, env)
        assert nn.tower_height == 0
        pi, v = nn.engine.forward(d["boards"], d["deltas"], d["ndeltas"], d["to_play"])
        assert np.abs(pi - d["pi_f64"]).max() <= 1e-4 and np.abs(v - d["v_f64"]).max() <= 1e-4
        out = os.path.join(t, "resaved")
        ag.save_model(nn, out)
        for part in ("base", "value", "policy"):
            a = open(os.path.join(t, "weights", f"agz_{part}.bson"), "rb").read()
            b = open(os.path.join(out, "weights", f"agz_{part}.bson"), "rb").read()
            assert a == b, part
        nn.engine.close()
