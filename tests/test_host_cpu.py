"""CPU-side checks of the product package: C-ABI surface, parameter tree, host geometry."""
import ctypes
import importlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG


def test_capi_exports_every_declared_symbol(pkg):
    lib = pkg._lib.load()
    names = pkg._lib.exported_symbols()
    assert len(names) >= 27
    for n in names:
        assert hasattr(lib, n), n
    assert lib.hh_abi_version() == 3


@pytest.mark.parametrize("C", [32, 48])
def test_state_dict_keys_match_engine_and_reference_count(pkg, C):
    net = pkg.HigherHRNet(17, C)
    keys = list(net.state_dict().keys())
    assert len(keys) == 1810
    assert net.engine_param_names() == keys
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == {32: 28645331, 48: 63827139}[C]


def test_classifier_state_dict_and_flops(pkg):
    net = pkg.ClassificationHRNet(32, 1000)
    assert net.engine_param_names() == list(net.state_dict().keys())
    assert sum(p.numel() for p in net.parameters()) == 41232680
    assert abs(net.forward_flops(1, 224, 224) / 2e9 - 8.921) < 2e-3


def test_algorithmic_flops_match_survey(pkg):
    assert abs(pkg.HigherHRNet(17, 32).forward_flops(1, 512, 512) / 2e9 - 46.203) < 1e-3
    assert abs(pkg.HigherHRNet(17, 48).forward_flops(1, 640, 640) / 2e9 - 149.469) < 1e-3


def test_load_weights_is_strict(pkg):
    lib = pkg._lib.load()
    h = lib.hh_create(17, 32, 1)
    a = np.zeros((64, 3, 3, 3), np.float32)
    shape = (ctypes.c_int64 * 4)(64, 3, 3, 3)
    assert lib.hh_load_weights(h, b"backbone.conv1.weight", a.ctypes.data, shape, 4) == 0
    assert lib.hh_load_weights(h, b"backbone.convX.weight", a.ctypes.data, shape, 4) != 0
    assert b"unexpected key" in lib.hh_last_error()
    bad = (ctypes.c_int64 * 4)(64, 3, 3, 1)
    assert lib.hh_load_weights(h, b"backbone.conv1.weight", a.ctypes.data, bad, 4) != 0
    assert lib.hh_finalize(h) != 0 and b"never loaded" in lib.hh_last_error()
    lib.hh_destroy(h)
    assert not lib.hh_create(17, 33, 1) and not lib.hh_decoder_create(17, 64, 0.1, 1.0)


def test_no_cpu_fallback(pkg):
    net = pkg.HigherHRNet(17, 32).eval()
    with pytest.raises(pkg._lib.HHError):
        net(torch.zeros(1, 3, 64, 64))
    with pytest.raises(pkg._lib.HHError):  # the training forward runs on the HIP kernels as well: no CPU path either
        pkg.HigherHRNet(17, 32).train()(torch.zeros(1, 3, 64, 64))
    with pytest.raises(NotImplementedError):  # the raw engine entry point is inference only
        pkg.HigherHRNet(17, 32).train().forward_raw(torch.zeros(1, 3, 64, 64))
    with pytest.raises(pkg._lib.HHError):
        pkg.MPPEHeatmapParser(17).parse(torch.zeros(17, 64, 64), torch.zeros(17, 64, 64, 1))


def test_multi_scale_size_and_affine(pkg):
    tu = importlib.import_module(PKG + ".keypoints.transforms_utils")
    rows = json.load(open(os.path.join(GOLDEN, "multi_scale_size.json")))
    for r in rows:
        size, center, scale = tu.get_multi_scale_size(np.zeros((r["h"], r["w"], 3), np.uint8), r["input_size"], r["current_scale"], r["min_scale"])
        assert list(size) == r["size"] and list(center) == r["center"] and [float(scale[0]), float(scale[1])] == r["scale"]
    # forward and inverse matrices are inverses; centre maps to centre
    m = tu.affine_matrix((320, 240), (660.0, 480.0), (704, 512))
    mi = tu.affine_matrix((320, 240), (660.0, 480.0), (704, 512), inverse=True)
    assert np.allclose(np.vstack([m, [0, 0, 1]]) @ np.vstack([mi, [0, 0, 1]]), np.eye(3), atol=1e-12)
    assert np.allclose(m @ [320, 240, 1], [352, 256])
    # transform_coords agrees with the matrix
    res = importlib.import_module(PKG + ".keypoints.results")
    xy = np.array([[[10.25, 20.75], [352.0, 256.0]]], np.float32)
    out = res.transform_coords(xy, (320, 240), (660.0, 480.0), (704, 512))
    exp = (mi @ np.array([[10.25, 20.75, 1], [352.0, 256.0, 1]]).T).T
    assert np.allclose(out[0], exp, atol=1e-4)


def test_affine_host_helpers_equal_the_oracle_restatement(pkg):
    """hh_get_affine_transform / hh_invert_affine / hh_transform_coords (C++ behind the C-ABI) against oracle/transforms.py (numpy),
    two independent statements of cv2.getAffineTransform's LU solve on the reference's float32 points, of the inversion
    cv2.warpAffine does, and of results.py:158-171 -- over every geometry of the multi_scale_size table, bit for bit.
    (Parity with cv2 itself is unpinned: no cv2 here.)"""
    from oracle import transforms as ot
    tu = importlib.import_module(PKG + ".keypoints.transforms_utils")
    res = importlib.import_module(PKG + ".keypoints.results")
    rows = json.load(open(os.path.join(GOLDEN, "multi_scale_size.json")))
    rng = np.random.default_rng(0)
    for r in rows:
        size, center, scale = tuple(r["size"]), tuple(r["center"]), tuple(r["scale"])
        assert ot.get_multi_scale_size((r["h"], r["w"]), r["input_size"], r["current_scale"], r["min_scale"]) == (size, center, scale)
        fwd = ot.get_affine_transform(center, scale, 0, size)
        assert np.array_equal(tu.affine_matrix(center, scale, size), fwd), r
        assert np.array_equal(tu.affine_matrix(center, scale, size, inverse=True), ot.get_affine_transform(center, scale, 0, size, inverse=True)), r
        assert np.array_equal(tu.dst_to_src_matrix(center, scale, size), ot.invert_affine(fwd)), r
        pts = np.concatenate([rng.uniform(0, max(size), (17, 2)), rng.uniform(0, 1, (17, 2))], 1).astype(np.float32)  # (x, y, score, tag)
        got = res.transform_coords(pts[:, :2], center, scale, size)
        ref = ot.transform_coords(pts, center, scale, size)
        assert got.dtype == np.float32 and np.array_equal(got, ref[:, :2]), r
        # the closed form the solve approximates: an isotropic scale about the centres (scale_h plays no role)
        rr = scale[0] / size[0]
        assert np.allclose(got, (pts[:, :2] - [size[0] / 2, size[1] / 2]) * rr + center, atol=1e-3)


def test_parse_checkpoint_prefixes(pkg):
    mod = importlib.import_module(PKG + ".keypoints.model")
    out = mod.parse_checkpoint({"module.net.backbone.conv1.weight": 1, "_orig_mod.net.init_heatmaps_head.bias": 2})
    assert out == {"backbone.conv1.weight": 1, "init_heatmaps_head.bias": 2}


def test_load_checkpoint_accepts_trainer_and_bare_layouts(tmp_path):
    """base/model.py:167-175: weights under ["module"]["model"] (trainer checkpoint) or a bare state dict (the published
    pretrained file); DDP / torch.compile / wrapper prefixes stripped either way."""
    import types
    import torch
    from torch import nn
    mod = importlib.import_module(PKG + ".keypoints.model")
    src = nn.Conv2d(3, 4, 3)
    sd = {"module.net." + k: v.clone() for k, v in src.state_dict().items()}
    for name, payload in (("trainer.pt", {"module": {"model": sd, "optimizers": {}}, "epoch": 3}), ("bare.pt", sd)):
        path = str(tmp_path / name)
        torch.save(payload, path)
        dst = nn.Conv2d(3, 4, 3)
        mod.InferenceKeypointsModel.load_checkpoint(types.SimpleNamespace(net=dst), path)
        assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values())), name


def test_coco_result_packing_matches_reference_layout():
    """bin/eval.py:27-48: [x, y, 1] * K float64 keypoints, category 1, score = the person score as a float."""
    ev = importlib.import_module(PKG + ".keypoints.evaluation")
    coords = np.arange(2 * 17 * 2, dtype=np.float32).reshape(2, 17, 2) / 3
    scores = np.array([0.75, 0.125], np.float32)
    out = ev.pack_coco_results(139, coords, scores)
    assert [r["image_id"] for r in out] == [139, 139] and all(r["category_id"] == 1 for r in out)
    for p, r in enumerate(out):
        k = np.asarray(r["keypoints"]).reshape(17, 3)
        assert np.array_equal(k[:, 0], coords[p, :, 0].astype(np.float64)) and np.array_equal(k[:, 1], coords[p, :, 1].astype(np.float64))
        assert np.all(k[:, 2] == 1) and isinstance(r["score"], float) and r["score"] == float(scores[p])
    assert ev.pack_coco_results(1, np.zeros((0, 17, 2), np.float32), np.zeros((0,), np.float32)) == []
    assert ev.image_id_from_path("/data/COCO/images/val2017/000000000139.jpg") == 139


def test_target_generators_properties():
    """HeatmapGenerator / JointsGenerator (datasets/coco.py:76-137) restated from the text: peak 1.0 at the joint,
    max-composition of overlapping blobs, clipping at the border, invisible and out-of-bounds joints dropped."""
    tg = importlib.import_module(PKG + ".keypoints.targets")
    raw = np.zeros((3, 17, 3))
    raw[0, 0] = (10.7, 20.2, 2); raw[0, 1] = (-3, 5, 2); raw[0, 2] = (5, 64, 1); raw[0, 3] = (63, 0, 1)
    raw[1, 0] = (12, 20, 1)
    j = tg.JointsGenerator(64)(raw)
    assert j.dtype == np.int32 and j.shape == (2, 17, 3)  # the all-zero third person is dropped
    assert j[0, 0].tolist() == [10, 20, 1] and j[0, 1].tolist() == [0, 0, 0] and j[0, 2].tolist() == [0, 0, 0] and j[0, 3].tolist() == [63, 0, 1]
    hm = tg.HeatmapGenerator(17, 64, 2)(j)
    assert hm.shape == (17, 64, 64) and hm.dtype == np.float32 and hm.max() == 1.0
    assert hm[0, 20, 10] == 1.0 and hm[0, 20, 12] == 1.0  # two people, same joint type: max, not sum
    assert abs(hm[0, 20, 11] - np.exp(-1 / 8)) < 1e-6 and hm[1].max() == 0 and hm[3, 0, 63] == 1.0
    assert hm[0, 20, 10 - 8] == 0 and hm[0, 20, 10 - 7] > 0  # support is 6*sigma+3 = 15 pixels
    lossmod = importlib.import_module(PKG + ".keypoints.loss")
    packed, counts = lossmod.pack_joints([j, np.zeros((0, 17, 3), np.int32)], 17, 64, 64)
    assert packed.shape == (2, 2, 17, 3) and counts.tolist() == [2, 0] and np.array_equal(packed[0], j)
    with pytest.raises(IndexError):
        lossmod.pack_joints([np.array([[[64, 0, 1]] * 17])], 17, 64, 64)


def _gt_person(img_id, ann_id, kpts, area, iscrowd=0):
    k = np.asarray(kpts, np.float64)
    x, y = k[:, 0], k[:, 1]
    return {"image_id": img_id, "id": ann_id, "category_id": 1, "iscrowd": iscrowd, "area": area,
            "num_keypoints": int((k[:, 2] > 0).sum()), "keypoints": k.reshape(-1).tolist(),
            "bbox": [float(x.min()), float(y.min()), float(x.max() - x.min()), float(y.max() - y.min())]}


def test_coco_oks_evaluator_hand_cases():
    """OKS / AP restated from the published pycocotools algorithm (package absent here: hand-derived expectations)."""
    ce = importlib.import_module(PKG + ".keypoints.coco_eval")
    ev_mod = importlib.import_module(PKG + ".keypoints.evaluation")
    rs = np.random.RandomState(0)
    gts, dets = [], []
    for img in (1, 2, 3):
        for p in range(2):
            k = np.concatenate([rs.uniform(50, 300, (17, 2)), np.full((17, 1), 2.0)], 1)
            gts.append(_gt_person(img, 10 * img + p, k, area=150.0 ** 2))
            dets += ev_mod.pack_coco_results(img, k[None, :, :2].astype(np.float32), np.array([0.9 - 0.1 * p], np.float32))
    ev = ce.COCOKeypointsEval(gts, dets)
    st = ev.evaluate()
    assert np.allclose(st[[0, 1, 2, 4, 5, 6, 7, 9]], 1.0) and st[3] == -1 and st[8] == -1  # all "large": medium has no gt
    # a known OKS: every joint displaced by d pixels -> OKS = mean(exp(-d^2 / (2 (2 sigma)^2 area)))
    d, area = 12.0, 150.0 ** 2
    shifted = []
    for g in gts:
        k = np.asarray(g["keypoints"]).reshape(17, 3)
        shifted.append({"image_id": g["image_id"], "category_id": 1, "score": 0.5,
                        "keypoints": np.concatenate([k[:, :2] + [d, 0.0], np.ones((17, 1))], 1).reshape(-1).tolist()})
    ev2 = ce.COCOKeypointsEval(gts, shifted)
    oks = ev2._oks(1)
    expect = np.mean(np.exp(-(d * d) / ((2 * ce.KPT_OKS_SIGMAS) ** 2) / (area + np.spacing(1)) / 2))
    assert abs(oks[0, 0] - expect) < 1e-12 or abs(oks[1, 1] - expect) < 1e-12
    st2 = ev2.evaluate()
    n_thr = int((ce.IOU_THRS <= expect + 1e-12).sum())  # matched at every threshold <= OKS, unmatched above
    assert abs(st2[0] - n_thr / 10.0) < 1e-9 and abs(st2[5] - n_thr / 10.0) < 1e-9
    # false positives ranked above the true ones halve the precision at every recall level
    fp = [dict(x, score=0.99, keypoints=(np.asarray(x["keypoints"]) + 1000).tolist()) for x in dets]
    st3 = ce.COCOKeypointsEval(gts, dets + fp).evaluate()
    assert abs(st3[0] - 0.5) < 0.02 and abs(st3[5] - 1.0) < 1e-9
    # crowd ground truth is ignored (its detection is neither TP nor FP); more than 20 detections per image are cut
    crowd = [dict(g, iscrowd=1) if i == 0 else g for i, g in enumerate(gts)]
    st4 = ce.COCOKeypointsEval(crowd, dets).evaluate()
    assert abs(st4[0] - 1.0) < 1e-9
    many = dets + [dict(dets[0], score=0.01 * i) for i in range(1, 40)]
    assert len(ce.COCOKeypointsEval(gts, many)._oks(1)) == 20


def test_multi_lane_schedule_has_no_unordered_hazard(pkg):
    """hh_debug_check_plan replays the fork/join/dependency edges of the multi-stream schedule with vector clocks: every
    RAW / WAR / WAW pair of ops on different lanes must be ordered (it reports the transition-conv race this repo once had)."""
    lib = pkg._lib.load()
    for net in (pkg.HigherHRNet(17, 32), pkg.HigherHRNet(17, 48), pkg.HigherHRNet(5, 32), pkg.ClassificationHRNet(32, 10)):
        assert lib.hh_debug_check_plan(net._h) == 0, lib.hh_last_error().decode()
    # the plan variants behind the engine switches (read once, in hh_create, into the handle) and the fp8 plan
    import os
    for env in ({"HH_NO_FUSION_MERGE": "1"}, {"HH_FULL_JOIN": "1"}, {"HH_FULL_JOIN": "1", "HH_NO_FUSION_MERGE": "1"}, {"HH_NO_STEM_FUSED": "1"}, {"HH_NO_JUNC_PAIR": "1"}, {"HH_NO_JUNC_PAIR": "1", "HH_FULL_JOIN": "1"}, {"HH_NO_HEAD_FOLD": "1"}, {"HH_BB32": "tile", "HH_NO_BB64": "1"}, {"HH_NO_CONV_DB": "1"}, {"HH_KEEP_WAITS": "1"}):
        os.environ.update(env)
        try:
            nets = (pkg.HigherHRNet(17, 32), pkg.HigherHRNet(17, 48))
        finally:
            for k in env:
                del os.environ[k]
        for net in nets:
            assert lib.hh_debug_check_plan(net._h) == 0, (env, lib.hh_last_error().decode())
    assert lib.hh_debug_check_plan(pkg.HigherHRNet(17, 48, dtype="fp8")._h) == 0, lib.hh_last_error().decode()


def test_keypoints_model_init_weights_and_wrappers():
    """KeypointsModel.init_weights (keypoints/model.py:19-34): conv / transposed-conv weights ~ N(0, 0.001) with zero bias,
    BatchNorm weight 1 / bias 0; init_pretrained_weights keeps the names that exist and drops the rest (base/model.py:101-123)."""
    import torch
    from torch import nn
    pkg = importlib.import_module(PKG)
    KeypointsModel = importlib.import_module(PKG + ".keypoints.model").KeypointsModel
    net = pkg.HigherHRNet(17, 32)
    model = KeypointsModel(net)
    torch.manual_seed(0)
    model.init_weights()
    n_conv = 0
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            n_conv += 1
            if m.weight.numel() > 4000:
                assert abs(m.weight.std().item() - 1e-3) < 1.5e-4 and abs(m.weight.mean().item()) < 1e-4
            assert m.bias is None or (m.bias == 0).all()
        elif isinstance(m, nn.BatchNorm2d):
            assert (m.weight == 1).all() and (m.bias == 0).all()
    assert n_conv == 303 and net._dirty  # 302 convs + the transposed conv
    sd = {("module.net." + k): (v + 1 if v.dtype.is_floating_point else v) for k, v in list(net.state_dict().items())[:7]}
    sd["module.net.not_a_layer.weight"] = torch.zeros(3)
    first = next(iter(net.state_dict().values())).clone()
    model.init_pretrained_weights(sd)
    assert torch.equal(next(iter(net.state_dict().values())), first + 1)
    assert set(model.state_dict()) == set(net.state_dict()) and model.device.type == "cpu"
    model.freeze()
    assert not any(p.requires_grad for p in net.parameters())


def test_e4m3_codec_matches_torch_float8(pkg):
    """The fp8 path's host quantiser (weights -> OCP e4m3fn) against torch's float8_e4m3fn conversion: all 256 codes decode
    alike, and random / boundary values inside the finite range encode to the same byte (round to nearest even); beyond
    +-448 this codec saturates where torch produces NaN."""
    import torch
    lib = pkg._lib.load()
    codes = np.arange(256, dtype=np.uint8)
    dec = np.empty(256, np.float32)
    lib.hh_e4m3_decode(codes.ctypes.data, 256, dec.ctypes.data)
    ref = torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(np.isnan(dec), np.isnan(ref)) and np.array_equal(dec[~np.isnan(ref)], ref[~np.isnan(ref)])
    rs = np.random.RandomState(0)
    finite = np.sort(np.unique(np.abs(ref[~np.isnan(ref)])))
    mids = (finite[:-1] + finite[1:]) / 2  # exact ties between neighbouring codes
    x = np.concatenate([rs.uniform(-448, 448, 20000), rs.normal(0, 1, 20000), rs.normal(0, 0.01, 20000), finite, -finite, mids, -mids,
                        np.nextafter(mids.astype(np.float32), np.float32(0)), np.nextafter(mids.astype(np.float32), np.float32(1e9)),
                        [0.0, -0.0, 1e-12, 447.9, 448.0, 463.9]]).astype(np.float32)
    enc = np.empty(x.size, np.uint8)
    lib.hh_e4m3_encode(x.ctypes.data, x.size, enc.ctypes.data)
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(enc, want), np.nonzero(enc != want)[0][:10]
    big = np.array([464.0, 1e6, -1e6, np.inf, -np.inf, np.nan], np.float32)
    out = np.empty(big.size, np.uint8)
    lib.hh_e4m3_encode(big.ctypes.data, big.size, out.ctypes.data)
    assert out.tolist() == [0x7e, 0x7e, 0xfe, 0x7e, 0xfe, 0x7f]


def test_coco_result_packing_equals_the_reference_run():
    """a19: tests/golden/eval_packing.json is what the REFERENCE's `evaluate_dataset` (src/keypoints/bin/eval.py:18-49, imported
    and run by tools/make_golden.py on a fake model / dataset) packed for the inputs in eval_packing_inputs.npz; this repo's
    packing must produce the same list -- ids from the zero-padded stems, [x, y, 1] * 17 as float64 of the float32 coordinates,
    python-float scores -- entry for entry, value for value."""
    ev = importlib.import_module(PKG + ".keypoints.evaluation")
    want = json.load(open(os.path.join(GOLDEN, "eval_packing.json")))
    d = np.load(os.path.join(GOLDEN, "eval_packing_inputs.npz"))
    got = []
    for i, stem in enumerate(d["stems"]):
        got += ev.pack_coco_results(ev.image_id_from_path(f"/data/COCO/images/val2017/{stem}.jpg"), d[f"coords{i}"], d[f"scores{i}"])
    assert len(got) == len(want) == 34
    assert json.loads(json.dumps(got)) == want
    assert [r["image_id"] for r in got][:4] == [139, 139, 139, 785] and got[-1]["image_id"] == 100000000001
    assert all(isinstance(r["score"], float) and len(r["keypoints"]) == 51 and r["keypoints"][2::3] == [1.0] * 17 for r in got)


def test_hand_counted_lds_waits_have_no_scalar_load_in_flight():
    """basicblock_fused_pc.hip / stem_fused.hip wait for their asm `ds_read_b128` fragments with `s_waitcnt lgkmcnt(N > 0)`, which is
    only sound while no scalar memory load (same counter, out-of-order return) is in flight at those waits.  The source keeps
    the kernel-argument reads outside the tile loops; tools/check_lds_wait_isa.py verifies on the emitted gfx950 assembly that the
    compiler kept them there."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(GOLDEN.rstrip("/")).rsplit("/tests", 1)[0]
    out = subprocess.run([_sys.executable, os.path.join(root, "tools", "check_lds_wait_isa.py"),
                          os.path.join(root, PKG, "csrc", "basicblock_fused_pc.hip"), os.path.join(root, PKG, "csrc", "stem_fused.hip")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "222 partial" in out.stdout or "partial lgkmcnt waits, 0 with" in out.stdout
