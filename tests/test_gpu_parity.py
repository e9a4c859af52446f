"""-m gpu: the HIP path (through the C-ABI) against the oracle and the golden vectors."""
import importlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, PKG
from oracle import decode as orc
from oracle import forward as ofw

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# bf16 activations + bf16 weights through ~60 sequential layers, fp32 accumulate.
# Stated tolerance (north_star "heatmaps/tags within a stated fp tolerance"):
#   max |engine - fp32 reference| <= 5e-2 * max |reference|, and rms error <= 2e-2 * rms(reference)
# (measured on the seeded W32 net: rms error grows from 0.2 % after the stem to ~1 % after stage 3)
TOL_MAX, TOL_RMS = 5e-2, 2e-2


def _net(pkg, C, seed):
    net = pkg.HigherHRNet(17, C)
    sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, seed)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    return net.to(DEV).eval(), sd


def _close(got, ref, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, what
    emax = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)
    erms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref**2).mean()), 1e-6)
    assert emax <= TOL_MAX and erms <= TOL_RMS, f"{what}: max {emax:.4f} rms {erms:.4f}"
    return emax, erms


def _same_bits(a, b, what):
    """Bit equality with a message that says WHERE two tensors differ (round 2 lost the text of a one-off failure)."""
    if torch.equal(a, b):
        return
    d = (a != b) | (a.isnan() != b.isnan())
    idx = d.nonzero()
    first = tuple(idx[0].tolist())
    raise AssertionError(f"{what}: {int(d.sum())} of {d.numel()} values differ, first at {first}: {a[first].item()!r} vs {b[first].item()!r}; "
                         f"NaNs {int(a.isnan().sum())} / {int(b.isnan().sum())}; differing index range {idx.min(0).values.tolist()} .. {idx.max(0).values.tolist()}")


def _switch_env(env):
    """Context: HH_* plan switches are read once in hh_create -- set them around the construction of a net only."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        os.environ.update(env)
        try:
            yield
        finally:
            for k in env:
                del os.environ[k]
    return cm()


def test_native_library_is_the_one_running(pkg):
    assert torch.cuda.is_available()
    lib = pkg._lib.load()
    assert os.path.samefile(lib._name, os.path.join(os.path.dirname(pkg.__file__), "csrc", "libhhrnet.so"))


def test_forward_with_taps_vs_reference_golden(pkg, net_golden):
    net, _ = _net(pkg, 32, 0)
    net.set_taps(True)
    x = torch.from_numpy(pkg.synth.synth_images(1, 64, 64, 0)).to(DEV)
    hms, tags = net(x)
    torch.cuda.synchronize()
    taps = net.read_taps()
    n = 0
    for k in net_golden.files:
        if k.startswith("w32_64/tap/"):
            name = k.split("/tap/")[1]
            if name == "deconv#1":
                continue
            _close(taps[name], net_golden[k], name)
            n += 1
    assert n >= 60
    _close(hms[0].cpu().numpy(), net_golden["w32_64/hm_q"], "hm_q")
    _close(hms[1].cpu().numpy(), net_golden["w32_64/hm_h"], "hm_h")
    _close(tags.cpu().numpy(), net_golden["w32_64/tags"], "tags")


@pytest.mark.parametrize("tag,C,B,H,W,seed", [("w32_128", 32, 2, 128, 128, 1), ("w32_96x160", 32, 1, 96, 160, 2), ("w48_64", 48, 1, 64, 64, 3)])
def test_forward_outputs_vs_reference_golden(pkg, net_golden, tag, C, B, H, W, seed):
    net, _ = _net(pkg, C, seed)
    x = torch.from_numpy(pkg.synth.synth_images(B, H, W, seed)).to(DEV)
    for use_graph in (False, True, True):  # eager, capture, replay
        net.use_graph = use_graph
        hms, tags = net(x)
        _close(hms[0].cpu().numpy(), net_golden[f"{tag}/hm_q"], "hm_q")
        _close(hms[1].cpu().numpy(), net_golden[f"{tag}/hm_h"], "hm_h")
        _close(tags.cpu().numpy(), net_golden[f"{tag}/tags"], "tags")


def test_fused_32_channel_block_both_forms(pkg, net_golden):
    """The 32-channel BasicBlock has two fused kernels: basicblock_fused_pc.hip (producer / consumer waves, the default) and
    basicblock_fused.hip (HH_BB32=tile).  Both meet the stated tolerance against the reference golden, differ from each other
    only by bf16 rounding of different summation orders, and the stand-alone A/B entry agrees on ragged shapes."""
    x = torch.from_numpy(pkg.synth.synth_images(2, 128, 128, 1)).to(DEV)
    pc_net, _ = _net(pkg, 32, 1)
    os.environ["HH_BB32"] = "tile"
    try:
        tile_net, _ = _net(pkg, 32, 1)
    finally:
        del os.environ["HH_BB32"]
    for net in (pc_net, tile_net):
        hms, tags = net(x)
        _close(hms[0].cpu().numpy(), net_golden["w32_128/hm_q"], "hm_q")
        _close(hms[1].cpu().numpy(), net_golden["w32_128/hm_h"], "hm_h")
        _close(tags.cpu().numpy(), net_golden["w32_128/tags"], "tags")
    a, b = pc_net.forward_raw(x), tile_net.forward_raw(x)
    assert not torch.equal(a[0], b[0])  # two kernels really ran
    for u, v in zip(a, b):
        assert (u - v).abs().max().item() <= 4e-2 * v.abs().max().item() and (u - v).pow(2).mean().sqrt().item() <= 2e-2 * v.pow(2).mean().sqrt().item()
    # one block, random input in +-[0.5, 4), weights +-[0.008, 0.03): outputs up to ~16, so one bf16 ulp is 0.0625
    lib = pkg._lib.load()
    import ctypes as C
    for (B, H, W) in [(1, 14, 32), (2, 30, 44), (3, 61, 77), (2, 128, 128)]:
        md, m0, m1 = C.c_float(), C.c_float(), C.c_float()
        pkg._lib.check(lib.hh_debug_bb_compare(B, H, W, 1, C.byref(md), C.byref(m0), C.byref(m1)))
        assert md.value <= 0.0625, (B, H, W, md.value)


def test_head_folded_into_the_transposed_conv(pkg, net_golden):
    """bf16 handles fold init_heatmaps_head into the transposed conv (higher_hrnet.py:52,70-74: the concat followed by a linear op is
    linear in the features; the head's bias rides on a constant-one channel, which is zero outside the image like every other
    channel, so the borders are exact).  Both plans meet the golden tolerance; the quarter-res outputs -- the same 1x1 launch either
    way -- are bit-identical; the half-res output differs only by the bf16 rounding of the intermediate the folded plan never forms;
    border rows / columns included on a ragged shape; the algorithmic FLOP count is the reference's either way."""
    with _switch_env({"HH_NO_HEAD_FOLD": "1"}):
        plain, _ = _net(pkg, 32, 1)
    fold, _ = _net(pkg, 32, 1)
    assert plain.forward_flops(1, 512, 512) == fold.forward_flops(1, 512, 512) == 2 * 46.2034e9 or abs(fold.forward_flops(1, 512, 512) / 2e9 - 46.2034) < 1e-3
    x = torch.from_numpy(pkg.synth.synth_images(2, 128, 128, 1)).to(DEV)
    for net in (fold, plain):
        hms, tags = net(x)
        _close(hms[0].cpu().numpy(), net_golden["w32_128/hm_q"], "hm_q")
        _close(hms[1].cpu().numpy(), net_golden["w32_128/hm_h"], "hm_h")
        _close(tags.cpu().numpy(), net_golden["w32_128/tags"], "tags")
    for shape in ((2, 128, 128), (3, 96, 160), (1, 32, 64)):
        x2 = torch.from_numpy(pkg.synth.synth_images(*shape, 4)).to(DEV)
        a, b = fold.forward_raw(x2), plain.forward_raw(x2)
        _same_bits(a[0], b[0], f"init_heatmaps {shape}")
        d = (a[1] - b[1]).abs()
        # (two bf16 paths through the four residual units of the deconv head: measured 1.0 % of max, 0.52 % rms)
        assert d.max().item() <= 4e-2 * b[1].abs().max().item() and d.pow(2).mean().sqrt().item() <= 1e-2 * b[1].pow(2).mean().sqrt().item(), shape
        edge = torch.cat([d[..., 0, :].reshape(-1), d[..., -1, :].reshape(-1), d[..., :, 0].reshape(-1), d[..., :, -1].reshape(-1)])
        assert edge.max().item() <= 4e-2 * b[1].abs().max().item(), shape  # (a wrong border term would be O(bias), far above bf16 noise)


def test_fused_stem_matches_two_launches(pkg, net_golden):
    """stem_fused.hip (both stem convs in one kernel, the default) against stem_conv.hip + a conv launch (HH_NO_STEM_FUSED=1): the
    stem tap and the outputs meet the golden tolerance either way, on a ragged batch / shape too the two paths agree to bf16 noise."""
    os.environ["HH_NO_STEM_FUSED"] = "1"
    try:
        two, _ = _net(pkg, 32, 1)
    finally:
        del os.environ["HH_NO_STEM_FUSED"]
    one, _ = _net(pkg, 32, 1)
    x = torch.from_numpy(pkg.synth.synth_images(2, 128, 128, 1)).to(DEV)
    for net in (one, two):
        hms, tags = net(x)
        _close(hms[0].cpu().numpy(), net_golden["w32_128/hm_q"], "hm_q")
        _close(hms[1].cpu().numpy(), net_golden["w32_128/hm_h"], "hm_h")
    for shape in ((3, 96, 160), (1, 32, 64), (5, 64, 32)):
        x2 = torch.from_numpy(pkg.synth.synth_images(*shape, 2)).to(DEV)
        a, b = one.forward_raw(x2), two.forward_raw(x2)
        for u, v in zip(a, b):
            assert (u - v).abs().max().item() <= 4e-2 * v.abs().max().item() and (u - v).pow(2).mean().sqrt().item() <= 2e-2 * v.pow(2).mean().sqrt().item(), shape


def test_schedule_and_fusion_switches(pkg, net_golden):
    """The plan variants behind the engine switches: all-to-all joins (HH_FULL_JOIN=1) give the default plan's bits (same kernels,
    other edges), and so do stage-0 junctions that each store their y (HH_NO_JUNC_PAIR=1: the pair mode makes the previous
    unit's y again with the same arithmetic and the same bf16 rounding); one launch per summed stride-2 conv (HH_NO_FUSION_MERGE=1) rounds the partial sums to bf16 between the launches
    and so differs from the merged conv by bf16 noise only, and so do the 128- / 256-channel 3x3 convs on the single-buffer 32-channel-chunk
    instantiations (HH_NO_CONV_DB=1: the double-buffered form walks 16-channel chunks, another order of the fp32 additions); all meet
    the golden tolerance."""
    x = torch.from_numpy(pkg.synth.synth_images(2, 128, 128, 1)).to(DEV)
    base, _ = _net(pkg, 32, 1)
    ref = [t.clone() for t in base.forward_raw(x)]
    # (round 4) the fused 32-channel block tiles the batch as ONE tall image, two zero rows between the images (HH_BB_TALL=always: here too,
    # where it needs more tiles than the per-image layout and is not chosen; HH_NO_BB_TALL=1: never): other tiles, the same sums
    for env, exact in ((("HH_FULL_JOIN", "1"), True), (("HH_NO_FUSION_MERGE", "1"), False), (("HH_NO_JUNC_PAIR", "1"), True), (("HH_NO_CONV_DB", "1"), False),
                       (("HH_BB_TALL", "always"), True), (("HH_NO_BB_TALL", "1"), True),
                       (("HH_KEEP_WAITS", "1"), True), (("HH_EVENT_SYSTEM_FENCE", "1"), True)):  # (round 4) enqueue() drops the waits its vector clocks prove redundant: the same launches either way
        os.environ[env[0]] = env[1]
        try:
            net, _ = _net(pkg, 32, 1)
        finally:
            del os.environ[env[0]]
        hms, tags = net(x)
        _close(hms[0].cpu().numpy(), net_golden["w32_128/hm_q"], "hm_q")
        _close(hms[1].cpu().numpy(), net_golden["w32_128/hm_h"], "hm_h")
        got = net.forward_raw(x)
        if exact:
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), env
        else:
            for a, b in zip(got, ref):
                assert (a - b).abs().max().item() <= 4e-2 * b.abs().max().item() and (a - b).pow(2).mean().sqrt().item() <= 2e-2 * b.pow(2).mean().sqrt().item()
    # the tall layout where the default plan picks it (4 images of 512 x 512: 152 tiles of the 128-row maps instead of 160, 304 of the
    # 256-row maps of the deconv head instead of 304 + ...) and on a ragged shape, against the per-image layout: bit for bit
    for shape in ((4, 512, 512), (5, 352, 416)):
        xs = torch.from_numpy(pkg.synth.synth_images(*shape, 9)).to(DEV)
        tall = [t.clone() for t in base.forward_raw(xs)]
        os.environ["HH_NO_BB_TALL"] = "1"
        try:
            plain, _ = _net(pkg, 32, 1)
        finally:
            del os.environ["HH_NO_BB_TALL"]
        assert all(torch.equal(a, b) for a, b in zip(plain.forward_raw(xs), tall)), shape


def test_final_layer_in_the_last_block_epilogue(pkg, net_golden):
    """(round 4) DeconvHeatmapsHead.final_layer (1x1, 32 -> 17, higher_hrnet.py:38-44) runs inside the last residual block's kernel
    (bbpc_final_kernel: the block output never goes to HBM) unless HH_NO_FINAL_FUSE=1 or taps are on.  Same bf16 operands, same
    fp32 accumulation over the same 32 channels in the same two MFMA steps, the bias added last instead of first: the half-resolution
    heatmaps agree to fp32 rounding (and meet the golden tolerance either way), everything else bit for bit.  Shapes: the golden one,
    the tall layout's (4 x 512 x 512), ragged tiles, a single tile."""
    os.environ["HH_NO_FINAL_FUSE"] = "1"
    try:
        plain, _ = _net(pkg, 32, 1)
    finally:
        del os.environ["HH_NO_FINAL_FUSE"]
    fused, _ = _net(pkg, 32, 1)
    x = torch.from_numpy(pkg.synth.synth_images(2, 128, 128, 1)).to(DEV)
    for net in (fused, plain):
        hms, _tags = net(x)
        _close(hms[1].cpu().numpy(), net_golden["w32_128/hm_h"], "hm_h")
    worst = 0.0
    for i, shape in enumerate(((2, 128, 128), (4, 512, 512), (5, 352, 416), (3, 96, 160), (1, 32, 64), (1, 64, 32))):
        xs = torch.from_numpy(pkg.synth.synth_images(*shape, 40 + i)).to(DEV)
        a = [t.clone() for t in fused.forward_raw(xs)]
        b = [t.clone() for t in plain.forward_raw(xs)]
        assert torch.equal(a[0], b[0]), shape  # init heatmaps + tags: untouched by the fusion
        assert a[1].shape == b[1].shape and torch.isfinite(a[1]).all(), shape
        d = (a[1] - b[1]).abs().max().item() / b[1].abs().max().item()
        worst = max(worst, d)
        assert d <= 2e-6, (shape, d)  # fp32 rounding of (sum + bias) against (bias + sum)
        c = [t.clone() for t in fused.forward_raw(xs)]
        assert torch.equal(a[1], c[1]), shape  # and it is deterministic
    print(f"fused final layer vs its own launch: max |diff| / max |ref| = {worst:.2e}")


def test_forward_reads_no_unwritten_workspace(pkg):
    """Recycled device memory is not zero: with the workspace filled with NaN patterns at allocation (HH_POISON_WS=1) the forward
    must give the bits it gives on fresh memory, for growing and shrinking shapes on one handle."""
    shapes = [(1, 96, 160), (2, 128, 128), (1, 64, 64)]
    xs = [torch.from_numpy(pkg.synth.synth_images(b, h, w, 30 + i)).to(DEV) for i, (b, h, w) in enumerate(shapes)]
    clean, _ = _net(pkg, 32, 0)
    ref = [[t.clone() for t in clean.forward_raw(x)] for x in xs]
    os.environ["HH_POISON_WS"] = "1"
    try:
        for order in ((0, 1, 2), (1, 0, 2, 1)):
            net, _ = _net(pkg, 32, 0)
            for i in order:
                got = net.forward_raw(xs[i])
                assert all(torch.equal(a, b) for a, b in zip(got, ref[i])), (order, i)
    finally:
        del os.environ["HH_POISON_WS"]


def test_forward_does_not_depend_on_stale_lds(pkg):
    """LDS is not cleared between kernels: with NaN patterns left in every CU's 160 KB in front of every launch
    (HH_POISON_LDS=1, one lane) the forward must give the bits it gives otherwise -- at 512x512 with B = 1 and B = 4 (the shapes of
    the one unexplained round-2 failure), on ragged shapes, and for every plan variant."""
    cases = [(1, 512, 512), (4, 512, 512), (1, 96, 160), (3, 64, 96), (2, 128, 128)]
    xs = [torch.from_numpy(pkg.synth.synth_images(b, h, w, 40 + i)).to(DEV) for i, (b, h, w) in enumerate(cases)]
    variants = [{}, {"HH_BB32": "tile"}, {"HH_NO_BB64": "1"}, {"HH_NO_HEAD_FOLD": "1"}, {"HH_NO_STEM_FUSED": "1"}, {"HH_NO_JUNC_PAIR": "1"},
                {"HH_NO_FUSION_MERGE": "1"}]
    for env in variants:
        with _switch_env(env):
            clean, _ = _net(pkg, 32, 0)
        with _switch_env(dict(env, HH_POISON_LDS="1")):
            dirty, _ = _net(pkg, 32, 0)
        for x, shape in zip(xs if not env else xs[:3], cases):
            ref = [t.clone() for t in clean.forward_raw(x)]
            got = dirty.forward_raw(x)
            for a, b, name in zip(got, ref, ("init_heatmaps", "deconv_heatmaps")):
                _same_bits(a, b, f"{env} {shape} {name} with poisoned LDS")


def test_many_live_handles_interleaved_forwards_stay_bit_exact(pkg):
    """Cross-handle state (round 2's open item): 17 handles over the plan switches stay ALIVE in one process, and B = 1 / B = 4
    forwards at 512x512 are interleaved over all of them for 50 rounds.  Every bit-exact variant must keep returning the bits a
    fresh default handle gave before the others existed; the two variants that round differently (tile-form block, unmerged
    fusion convs) must each keep returning their own first result."""
    x1 = torch.from_numpy(pkg.synth.synth_images(1, 512, 512, 7)).to(DEV)
    xb = torch.from_numpy(pkg.synth.synth_images(4, 512, 512, 8)).to(DEV)
    xb[0] = x1[0]
    xb[3] = x1[0]
    first, _ = _net(pkg, 32, 0)
    ref1 = [t.clone() for t in first.forward_raw(x1)]
    ref4 = [t.clone() for t in first.forward_raw(xb)]
    _same_bits(ref4[0][0], ref1[0][0], "fresh handle: init_heatmaps slot 0 vs batch of 1")
    _same_bits(ref4[1][3], ref1[1][0], "fresh handle: deconv_heatmaps slot 3 vs batch of 1")
    exact = [{}, {"HH_FULL_JOIN": "1"}, {"HH_NO_JUNC_PAIR": "1"}, {"HH_FULL_JOIN": "1", "HH_NO_JUNC_PAIR": "1"}, {}, {}]
    own = [{"HH_NO_STEM_FUSED": "1"}, {"HH_NO_STEM_FUSED": "1", "HH_NO_JUNC_PAIR": "1"}, {"HH_BB32": "tile"}, {"HH_NO_FUSION_MERGE": "1"}, {"HH_NO_BB64": "1"}, {"HH_NO_HEAD_FOLD": "1"}, {"HH_NO_HEAD_FOLD": "1", "HH_FULL_JOIN": "1"},
           {"HH_BB32": "tile", "HH_FULL_JOIN": "1"}, {"HH_NO_BB64": "1", "HH_NO_FUSION_MERGE": "1"}, {"HH_NO_CONV_DB": "1"}]
    nets = []
    for env in exact + own:
        with _switch_env(env):
            nets.append((env, _net(pkg, 32, 0)[0]))
    other = pkg.HigherHRNet(17, 48)  # a different architecture between them: other weights, other workspace
    other.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 3)) for k, v in other.state_dict().items()})
    other.to(DEV).eval()
    xo = torch.from_numpy(pkg.synth.synth_images(2, 128, 192, 5)).to(DEV)
    refo = [t.clone() for t in other.forward_raw(xo)]
    own_ref = {}
    order = list(range(len(nets)))
    rng = np.random.default_rng(0)
    for rnd in range(50):
        rng.shuffle(order)
        for i in order:
            env, net = nets[i]
            for x, ref, nm in ((x1, ref1, "B=1"), (xb, ref4, "B=4")) if (rnd + i) % 2 else ((xb, ref4, "B=4"), (x1, ref1, "B=1")):
                got = net.forward_raw(x)
                if i >= len(exact):
                    ref = own_ref.setdefault((i, nm), [t.clone() for t in got])
                for a, b, name in zip(got, ref, ("init_heatmaps", "deconv_heatmaps")):
                    _same_bits(a, b, f"round {rnd} handle {i} {env} {nm} {name}")
        if rnd % 10 == 0:
            for a, b in zip(other.forward_raw(xo), refo):
                _same_bits(a, b, f"round {rnd} W48 handle")


def test_forward_full_size_samples_and_batch_consistency(pkg, net_golden):
    net, _ = _net(pkg, 32, 0)
    x1 = torch.from_numpy(pkg.synth.synth_images(1, 512, 512, 7)).to(DEV)
    hms, tags = net(x1)
    for name, t in (("hm_q", hms[0]), ("hm_h", hms[1]), ("tags", tags)):
        got = t.cpu().numpy().reshape(-1)[net_golden[f"w32_512/{name}_idx"]]
        _close(got, net_golden[f"w32_512/{name}_val"], name)
    # images are independent: the same image at batch slots 0 and 3 of a batch of 4 gives identical bits
    xb = torch.from_numpy(pkg.synth.synth_images(4, 512, 512, 8)).to(DEV)
    xb[0] = x1[0]
    xb[3] = x1[0]
    hb, tb = net(xb)
    _same_bits(hb[0][0], hms[0][0], "hm_q, slot 0 of a batch of 4 vs batch of 1")
    _same_bits(hb[1][3], hms[1][0], "hm_h, slot 3 of a batch of 4 vs batch of 1")
    _same_bits(tb[3], tags[0], "tags, slot 3 of a batch of 4 vs batch of 1")


def test_flip_tta_vs_reference_golden(pkg):
    g = np.load(os.path.join(GOLDEN, "flip_tta.npz"))
    net, _ = _net(pkg, 32, 0)
    model = pkg.InferenceKeypointsModel(net, use_flip=True, device=DEV)
    x = torch.from_numpy(pkg.synth.synth_images(1, 64, 64, 21)).to(DEV)
    hms, tags = model.forward_tta(x)
    _close(hms[0].cpu().numpy(), g["hm_q"], "hm_q")
    _close(hms[1].cpu().numpy(), g["hm_h"], "hm_h")
    _close(tags[0].cpu().numpy(), g["tags0"], "tags0")
    _close(tags[1].cpu().numpy(), g["tags1"], "tags1")


def _case_inputs(synth, m):
    return synth.synth_decode_maps(17, m["hq"], m["wq"], m["people"], seed=m["seed"], emb=m["emb"], **m["kwargs"])


def test_decode_bit_exact_vs_reference_golden(pkg, synth, decode_golden):
    meta, g = decode_golden
    for tag, m in meta.items():
        hm_q, hm_h, tags, _ = _case_inputs(synth, m)
        parser = pkg.MPPEHeatmapParser(17, m["max_people"], m["det_thr"], m["tag_thr"])
        t = lambda a: torch.from_numpy(a)[None].to(DEV)  # noqa: E731
        for name, (a, r) in {"joints": (1, 1), "joints_norefine": (1, 0), "joints_noadjust": (0, 1)}.items():
            out = parser.decode_batch_device(t(hm_q), t(hm_h), [t(x) for x in tags], adjust=bool(a), refine=bool(r))
            j, s = parser.to_lists(*out)[0]
            if m["has_ties"]:  # torch.topk's tie order is unspecified: the oracle's index-ascending rule is the contract
                rj, rs = orc.decode(hm_q, hm_h, tags, max_people=m["max_people"], det_thr=m["det_thr"], tag_thr=m["tag_thr"], adjust=a, refine=r)
            else:
                rj, rs = g[tag + "/" + name], g[tag + "/scores"]
            assert j.dtype == rj.dtype and j.shape == rj.shape and np.array_equal(j, rj), (tag, name)
            if name == "joints":
                assert s.dtype == rs.dtype and np.array_equal(s, rs), tag
        # the candidate lists themselves need every tile (the default skips tiles that cannot reach det_thr: same parse results)
        outd = [x.clone() for x in parser.decode_batch_device(t(hm_q), t(hm_h), [t(x) for x in tags])]
        parser.set_exact_topk(True)
        oute = parser.decode_batch_device(t(hm_q), t(hm_h), [t(x) for x in tags])
        n0 = int(outd[2][0])
        assert int(oute[2][0]) == n0 and torch.equal(outd[0][0, :n0], oute[0][0, :n0]) and torch.equal(outd[1][0, :n0], oute[1][0, :n0]) and torch.equal(outd[3], oute[3]), tag
        tk, ck, sk = parser.last_top_k(1, m["emb"])
        parser.set_exact_topk(False)
        full, tfull = orc.aggregate(hm_q, hm_h, tags)
        otk, ock, osk = orc.top_k(full, tfull, m["max_people"])
        assert np.array_equal(sk[0], osk) and np.array_equal(ck[0], ock) and np.array_equal(tk[0], otk), tag
        if not m["has_ties"]:
            pos = g[tag + "/scores_k"] > 0
            assert np.array_equal(sk[0][pos], g[tag + "/scores_k"][pos]) and np.array_equal(ck[0][pos], g[tag + "/coords_k"][pos])


def test_gpu_assignment_solver_vs_pinned_munkres(pkg):
    """The matcher's step machine alone (hh_debug_munkres: one wavefront, 32-bit zero / cover masks, per-column zero masks that keep
    step 4's row set up to date): the 60 matrices whose assignments munkres 1.1.4 itself wrote (tests/golden/munkres.npz; rectangular
    ones padded with zeros as munkres.pad_matrix does), then 300 seeded square problems with many equal entries -- the
    `round(dist) * 100 - score` costs of grouping.py:121-125 tie often, and WHICH zero is starred first is the contract -- against
    the C restatement that those goldens pin."""
    import ctypes as C
    lib = pkg._lib.load()

    def solve(cost):
        r, c = cost.shape
        n = max(r, c)
        sq = np.zeros((n, n), np.float64)
        sq[:r, :c] = cost
        star = np.full(n, -9, np.int32)
        assert lib.hh_debug_munkres(sq.ctypes.data_as(C.POINTER(C.c_double)), n, star.ctypes.data_as(C.POINTER(C.c_int32))) == 0, lib.hh_last_error().decode()
        return np.array([(i, star[i]) for i in range(r) if 0 <= star[i] < c], np.int32).reshape(-1, 2)

    d = np.load(os.path.join(GOLDEN, "munkres.npz"))
    for i in range(len(d.files) // 2):
        assert np.array_equal(solve(d[f"m{i}"]), d[f"r{i}"]), f"golden {i} {d[f'm{i}'].shape}"
    rng = np.random.default_rng(77)
    for t in range(300):
        n = int(rng.integers(1, 33))
        kind = t % 4
        if kind == 0:
            cost = rng.integers(0, 4, (n, n)).astype(np.float64) * 100.0 - rng.integers(0, 3, (n, n)) * 0.25  # few distinct values
        elif kind == 1:
            cost = np.round(rng.uniform(0, 3, (n, n))) * 100.0 - rng.uniform(0.05, 1.0, (n, 1)).astype(np.float32).astype(np.float64)
        elif kind == 2:
            cost = rng.uniform(0, 5, (n, n))
            cost[:, rng.integers(0, n, max(1, n // 3))] = 1e10  # the reference's padding columns (grouping.py:126-128)
        else:
            cost = np.zeros((n, n))  # everything ties
            cost[rng.integers(0, n, n), rng.integers(0, n, n)] = 1.0
        assert np.array_equal(solve(cost), orc.munkres(cost)), (t, n, kind)


def test_decode_fuzz_vs_oracle(pkg, synth):
    """Seeded sweep beside the 14 reference goldens: 48 constructed cases over ragged map sizes, 0..30 people (more candidates
    than `max_people` included), one and two embedding maps (flip TTA), both threshold pairs the reference uses (inference
    0.05 / 0.5, validation 0.1 / 1.0), max_people 5 / 20 / 30, noisy and crowded tags, adjust / refine on and off -- every case must
    equal the C oracle (the restatement the goldens pin) bit for bit, dtype included, in one batched call per shape."""
    rng = np.random.default_rng(20261005)
    shapes = [(20, 28), (24, 40), (32, 32), (40, 24), (48, 64), (64, 64)]  # (model inputs >= 80 px: above torch's small-size bilinear path)
    for si, (hq, wq) in enumerate(shapes):
        for emb in (1, 2):
            cases = []
            for c in range(4):
                people = int(rng.integers(0, 31)) if c else 0
                cases.append(dict(people=people, seed=int(rng.integers(1 << 20)), tag_noise=float(rng.choice([0.02, 0.05, 0.2])),
                                  spacing=float(rng.choice([0.6, 1.0, 1.7])), drop=float(rng.choice([0.0, 0.15, 0.5]))))
            maps = [synth.synth_decode_maps(17, hq, wq, cs["people"], seed=cs["seed"], emb=emb, tag_noise=cs["tag_noise"],
                                            tag_spacing=cs["spacing"], drop_prob=cs["drop"]) for cs in cases]
            hm_q = torch.from_numpy(np.stack([m[0] for m in maps])).to(DEV)
            hm_h = torch.from_numpy(np.stack([m[1] for m in maps])).to(DEV)
            tg = [torch.from_numpy(np.stack([m[2][e] for m in maps])).to(DEV) for e in range(emb)]
            det, tagt = ((0.05, 0.5), (0.1, 1.0))[(si + emb) % 2]
            mp = (30, 20, 5)[(si + emb) % 3]
            adj, ref = bool((si + emb) % 4 != 1), bool((si + emb) % 4 != 2)
            parser = pkg.MPPEHeatmapParser(17, mp, det, tagt)
            res = parser.to_lists(*parser.decode_batch_device(hm_q, hm_h, tg, adjust=adj, refine=ref))
            for b, m in enumerate(maps):
                rj, rs = orc.decode(m[0], m[1], m[2], max_people=mp, det_thr=det, tag_thr=tagt, adjust=int(adj), refine=int(ref))
                ctx = (hq, wq, emb, cases[b], det, tagt, mp, adj, ref)
                assert res[b][0].dtype == rj.dtype and res[b][0].shape == rj.shape and np.array_equal(res[b][0], rj), ctx
                assert res[b][1].dtype == rs.dtype and np.array_equal(res[b][1], rs), ctx


def test_parse_fullres_boundary_bit_exact(pkg, synth, decode_golden):
    """MPPEHeatmapParser.parse on explicit full-resolution maps (the reference's parser boundary)."""
    meta, g = decode_golden
    for tag in ("p3_160_e2", "p6_missing", "p0_160", "p10_512"):
        m = meta[tag]
        hm_q, hm_h, tags, _ = _case_inputs(synth, m)
        full, tfull = orc.aggregate(hm_q, hm_h, tags)
        parser = pkg.MPPEHeatmapParser(17, m["max_people"], m["det_thr"], m["tag_thr"])
        j, s = parser.parse(torch.from_numpy(full).to(DEV), torch.from_numpy(tfull).to(DEV))
        assert j.dtype == g[tag + "/joints"].dtype and np.array_equal(j, g[tag + "/joints"]) and s.dtype == g[tag + "/scores"].dtype and np.array_equal(s, g[tag + "/scores"]), tag
        tk, ck, sk = parser.top_k(torch.from_numpy(full).to(DEV), torch.from_numpy(tfull).to(DEV))
        pos = g[tag + "/scores_k"] > 0
        assert np.array_equal(sk[pos], g[tag + "/scores_k"][pos]) and np.array_equal(tk[pos], g[tag + "/tags_k"][pos])


def test_decode_batch_is_per_image_and_ragged(pkg, synth):
    """Full bench size (B=32, 512x512 model input): every image decodes as it does alone."""
    B = 32
    maps = [synth.synth_decode_maps(17, 128, 128, (b * 7) % 13, seed=100 + b) for b in range(B)]  # 0..12 people
    hm_q = torch.from_numpy(np.stack([m[0] for m in maps])).to(DEV)
    hm_h = torch.from_numpy(np.stack([m[1] for m in maps])).to(DEV)
    tg = torch.from_numpy(np.stack([m[2][0] for m in maps])).to(DEV)
    parser = pkg.MPPEHeatmapParser(17, 30, 0.05, 0.5)
    res = parser.to_lists(*parser.decode_batch_device(hm_q, hm_h, [tg]))
    for b in (0, 5, 13, 31):
        rj, rs = orc.decode(maps[b][0], maps[b][1], maps[b][2], max_people=30, det_thr=0.05, tag_thr=0.5)
        assert res[b][0].shape == rj.shape and np.array_equal(res[b][0], rj) and np.array_equal(res[b][1], rs), b
    # idempotence: decoding the same device buffers again gives the same bits
    res2 = parser.to_lists(*parser.decode_batch_device(hm_q, hm_h, [tg]))
    assert all(np.array_equal(a[0], b[0]) for a, b in zip(res, res2))


def test_end_to_end_model_call(pkg, synth):
    """InferenceKeypointsModel.__call__ on a raw uint8 image: decode of the engine's own maps == oracle decode."""
    net, _ = _net(pkg, 32, 0)
    model = pkg.InferenceKeypointsModel(net, det_thr=0.05, tag_thr=0.5, use_flip=True, input_size=256, device=DEV)
    img = np.random.RandomState(3).randint(0, 255, (200, 300, 3)).astype(np.uint8)
    res = model(img, None)
    assert model.model_input_shape == (256, 384)
    x, center, scale = model.prepare_input(img)
    hms, tags = model.forward_tta(x)
    rj, rs = orc.decode(hms[0][0].cpu().numpy(), hms[1][0].cpu().numpy(), [t[0].cpu().numpy() for t in tags], max_people=30, det_thr=0.05, tag_thr=0.5)
    assert res.kpts_scores.shape == rj.shape[:2] and np.array_equal(res.kpts_scores, rj[..., 2]) and np.array_equal(res.obj_scores, rs)
    exp = orc.transform_coords(rj[..., :2], center, scale, (384, 256)).reshape(rj.shape[0], 17, 2)
    assert np.array_equal(res.kpts_coords, exp)  # hh_transform_coords and the oracle restate the same LU solve: same bits


def test_classification_hrnet_cfg1_vs_reference_golden(pkg):
    """BASELINE.json configs[0] through the HIP engine: ClassificationHRNet-W32, one 224x224 image."""
    net = pkg.ClassificationHRNet(32, 1000)
    sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 11)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net.to(DEV).eval()
    x = torch.from_numpy(pkg.synth.synth_images(1, 224, 224, 11)).to(DEV)
    logits = net(x).cpu().numpy()
    ref = np.load(os.path.join(GOLDEN, "cls_forward.npz"))["logits"]
    _close(logits, ref, "logits")
    assert int(logits.argmax()) == int(ref.argmax())


def _decode_both(pkg, hm_q, hm_h, tags, K, maxp=30, det=0.05, tthr=0.5):
    parser = pkg.MPPEHeatmapParser(K, maxp, det, tthr)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))[None].to(DEV)  # noqa: E731
    j, s = parser.to_lists(*parser.decode_batch_device(t(hm_q), t(hm_h), [t(x) for x in tags]))[0]
    rj, rs = orc.decode(hm_q, hm_h, tags, max_people=maxp, det_thr=det, tag_thr=tthr)
    assert j.dtype == rj.dtype and j.shape == rj.shape and np.array_equal(j, rj) and s.dtype == rs.dtype and np.array_equal(s, rs)
    return j, s


def test_decode_degenerate_and_edge_inputs(pkg, synth):
    """Edge cases of the decode domain, each bit-exact against the oracle: empty maps, constant maps (every pixel a
    tied peak), negative-only maps, peaks on the image border, people caps 1 and 32, K != 17, non-square and large maps."""
    rs = np.random.RandomState(0)
    K, hq, wq = 17, 40, 40
    z = lambda *s: np.zeros(s, np.float32)  # noqa: E731
    # all zeros -> no candidate above det_thr -> fallback person (grouping.py:262-269)
    j, s = _decode_both(pkg, z(K, hq, wq), z(K, 2 * hq, 2 * wq), [z(K, hq, wq)], K)
    assert j.shape[0] == 1 and j.dtype == np.float64 and s.dtype == np.float64 and np.all(j[0, :, 2] == 0.01)  # float64, as the reference returns it
    # constant positive maps: every pixel is a tied local maximum above det_thr
    c = np.full((K, hq, wq), 0.5, np.float32)
    _decode_both(pkg, c, np.full((K, 2 * hq, 2 * wq), 0.5, np.float32), [z(K, hq, wq)], K)
    # strictly negative maps (no zeros among the peaks of a plateau-free random field)
    _decode_both(pkg, -rs.uniform(0.1, 1, (K, hq, wq)).astype(np.float32), -rs.uniform(0.1, 1, (K, 2 * hq, 2 * wq)).astype(np.float32),
                 [rs.randn(K, hq, wq).astype(np.float32)], K)
    # single peaks in the four corners / on the borders: adjust and refine clamp at the edges
    hm_q, hm_h = z(K, hq, wq), z(K, 2 * hq, 2 * wq)
    tg = z(K, hq, wq) + 1.0
    for k, (y, x) in enumerate([(0, 0), (0, wq - 1), (hq - 1, 0), (hq - 1, wq - 1), (0, 17), (hq - 1, 5), (9, 0), (30, wq - 1)]):
        hm_q[k, y, x] = 0.9
        hm_h[k, 2 * y, 2 * x] = 0.8
    _decode_both(pkg, hm_q, hm_h, [tg], K)
    # people caps and a non-COCO joint count (joints_order falls back to 0..K-1)
    for maxp in (1, 32):
        a, b, t, _ = synth.synth_decode_maps(K, 48, 48, 12, seed=50 + maxp)
        _decode_both(pkg, a, b, t, K, maxp=maxp)
    a, b, t, _ = synth.synth_decode_maps(5, 40, 56, 4, seed=60, emb=2)
    _decode_both(pkg, a, b, t, 5)
    # a large, non-square map: 768 x 1024 model input (tiles straddle the border in y: 768 = 12 tiles, 1024 = 16)
    a, b, t, _ = synth.synth_decode_maps(K, 192, 256, 6, seed=70)
    _decode_both(pkg, a, b, t, K)
    # 64 joints x 2 embedding dimensions: the matching kernel's candidate tables no longer fit its LDS staging (global path)
    a, b, t, _ = synth.synth_decode_maps(64, 24, 24, 3, seed=80, emb=2)
    _decode_both(pkg, a, b, t, 64)
    # 244 x 244: the last NMS tile of a row / column is 4 pixels wide, the corner tile 4 x 4 (fewer pixels than top-k entries)
    a, b, t, _ = synth.synth_decode_maps(K, 61, 61, 3, seed=71)
    _decode_both(pkg, a, b, t, K)


def test_error_paths_raise(pkg):
    net, _ = _net(pkg, 32, 0)
    with pytest.raises(pkg._lib.HHError):
        net(torch.zeros(1, 3, 100, 128, device=DEV))  # H not a multiple of 32
    parser = pkg.MPPEHeatmapParser(17, 30, 0.05, 0.5)
    with pytest.raises(pkg._lib.HHError):
        t = torch.zeros(1, 17, 40, 40, device=DEV)
        parser.decode_batch_device(t, torch.zeros(1, 17, 80, 80, device=DEV), [t] * 5)  # E > 4
    with pytest.raises(pkg._lib.HHError):
        pkg.MPPEHeatmapParser(17, 33, 0.05, 0.5)  # max_num_people > 32


def test_forward_odd_batch_and_repeatability(pkg):
    net, sd = _net(pkg, 32, 0)
    x = torch.from_numpy(pkg.synth.synth_images(3, 128, 192, 5)).to(DEV)
    h1, t1 = net(x)
    h2, t2 = net(x)
    assert torch.equal(h1[0], h2[0]) and torch.equal(h1[1], h2[1]) and torch.equal(t1, t2)  # deterministic
    with torch.no_grad():
        rh, rt = ofw.higher_hrnet(x.cpu(), sd, 17)
    _close(h1[1].cpu().numpy(), rh[1].numpy(), "hm_h")
    _close(t1.cpu().numpy(), rt.numpy(), "tags")


def test_gpu_preprocessing_matches_oracle_opencv_restatement(pkg):
    """hh_preprocess_u8 == Normalize(ToTensor(cv2.warpAffine(image, get_affine_transform(...)))) as oracle/transforms.py restates
    OpenCV 4.9's fixed-point warp (10-bit coordinates, 5-bit fractions, int16 weight table, (v + 2^14) >> 15), bit for bit; the
    warped uint8 image itself (hh_warp_affine_u8) too.  Parity with cv2 itself stays unpinned (no cv2 here)."""
    import importlib
    from oracle import transforms as ot
    tu = importlib.import_module(pkg.__name__ + ".keypoints.transforms_utils")
    net, _ = _net(pkg, 32, 0)
    model = pkg.InferenceKeypointsModel(net, input_size=256, device=DEV)
    for shape in ((200, 300, 3), (301, 177, 3), (256, 256, 3), (97, 411, 3), (480, 640, 3)):
        img = np.random.RandomState(shape[0]).randint(0, 256, shape).astype(np.uint8)
        x, center, scale = model.prepare_input(img)
        ref, resized, c2, s2 = ot.prepare_input(img, 256)
        assert tuple(center) == tuple(c2) and tuple(scale) == tuple(s2)
        assert x.shape[1:] == ref.shape
        assert np.array_equal(x[0].cpu().numpy(), ref), shape
        got_u8, c3, s3 = tu.resize_align_multi_scale(img, 256, 1, 1, DEV)
        assert np.array_equal(got_u8, resized), shape
    # a rotated / sheared / shifted warp with out-of-image regions on every side (the kernel takes any affine)
    img = np.random.RandomState(5).randint(0, 256, (90, 130, 3)).astype(np.uint8)
    for m in ([[0.8, 0.3, -12.5], [-0.25, 1.1, 20.25]], [[1.7, 0, -40], [0, 1.7, -33.3]], [[0.31, -0.05, 7], [0.02, 0.29, 3]]):
        assert np.array_equal(tu.warp_affine(img, np.array(m), (150, 110), DEV), ot.warp_affine(img, m, (150, 110))), m
    # hh_preprocess_u8_batch: raw images of different sizes (one aspect ratio -> one model-input shape) in one launch, descriptors
    # read from device memory: the same bits as image by image
    import ctypes as C
    mod = importlib.import_module(pkg.__name__ + ".keypoints.model")
    imgs = [np.random.RandomState(s).randint(0, 256, shp).astype(np.uint8) for s, shp in ((1, (200, 300, 3)), (2, (400, 600, 3)), (3, (100, 150, 3)))]
    singles = [model.prepare_input(im)[0] for im in imgs]
    assert len({tuple(t.shape) for t in singles}) == 1
    H, W = singles[0].shape[-2:]
    offs = np.cumsum([0] + [im.size for im in imgs])
    desc_off = (int(offs[-1]) + 63) // 64 * 64
    buf = np.zeros(desc_off + 64 * len(imgs), np.uint8)
    descs = buf[desc_off:].view(mod._IMAGE_DESC)
    for j, im in enumerate(imgs):
        buf[offs[j]:offs[j + 1]] = im.reshape(-1)
        descs[j] = (int(offs[j]), im.shape[0], im.shape[1], model._geometry(im)[3].reshape(6))
    raw = torch.from_numpy(buf).to(DEV)
    out = torch.empty((len(imgs), 3, H, W), device=DEV)
    lib = pkg._lib.load()
    fp = C.POINTER(C.c_float)
    pkg._lib.check(lib.hh_preprocess_u8_batch(raw.data_ptr(), raw.data_ptr() + desc_off, len(imgs), out.data_ptr(), H, W,
                                              tu.IMAGENET_MEAN.ctypes.data_as(fp), tu.IMAGENET_STD.ctypes.data_as(fp),
                                              torch.cuda.current_stream().cuda_stream))
    for j, t in enumerate(singles):
        assert torch.equal(out[j], t[0])


def test_resize_accumulate_and_multi_scale_extension(pkg):
    """hh_resize_accumulate is torch's bilinear for arbitrary ratios (checked bit-exactly against the oracle's pinned
    restatement), and the multi-scale test (BASELINE configs[3], W48) equals the same aggregation done with the oracle."""
    lib = pkg._lib.load()
    rs = np.random.RandomState(1)
    for (h, w, H, W) in [(40, 56, 80, 112), (80, 112, 40, 56), (17, 23, 100, 77), (64, 64, 64, 64)]:
        src = torch.from_numpy(rs.randn(2, 5, h, w).astype(np.float32)).to(DEV)
        dst = torch.full((2, 5, H, W), 7.0, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        pkg._lib.check(lib.hh_resize_accumulate(src.data_ptr(), src.stride(0), 2, 5, h, w, dst.data_ptr(), dst.stride(0), H, W, 0.5, 1, st))
        pkg._lib.check(lib.hh_resize_accumulate(src.data_ptr(), src.stride(0), 2, 5, h, w, dst.data_ptr(), dst.stride(0), H, W, 0.5, 0, st))
        ref = np.stack([orc.bilinear(src[b].cpu().numpy(), H, W) for b in range(2)])
        half = (np.float32(0.5) * ref).astype(np.float32)
        assert np.array_equal(dst.cpu().numpy(), (half + half).astype(np.float32))
    net, sd = _net(pkg, 48, 3)
    model = pkg.InferenceKeypointsModel(net, det_thr=0.05, tag_thr=0.5, use_flip=True, input_size=128, device=DEV)
    img = np.random.RandomState(9).randint(0, 255, (150, 220, 3)).astype(np.uint8)
    scales = (0.5, 1.0, 2.0)
    hms, tags, xb, center, scale = model.multi_scale_maps(img, scales)
    assert tuple(xb.shape[-2:]) == (128, 256)
    exp = [np.zeros(hms[0].shape[1:], np.float32), np.zeros(hms[1].shape[1:], np.float32)]
    for s in scales:
        x, _, _ = model.prepare_input_scaled(img, s, min(scales))
        assert tuple(x.shape[-2:]) == (int(128 * s), int(256 * s))
        h_s, _ = model.forward_tta(x)
        for st in range(2):
            r = orc.bilinear(h_s[st][0].cpu().numpy(), exp[st].shape[1], exp[st].shape[2])
            exp[st] = (exp[st] + np.float32(1.0 / 3.0) * r).astype(np.float32) if s != scales[0] else (np.float32(1.0 / 3.0) * r).astype(np.float32)
    for st in range(2):
        assert np.array_equal(hms[st][0].cpu().numpy(), exp[st])
    res = model.call_multi_scale(img, None, scales)
    rj, rsc = orc.decode(exp[0], exp[1], [t[0].cpu().numpy() for t in tags], max_people=30, det_thr=0.05, tag_thr=0.5)
    assert np.array_equal(res.kpts_scores, rj[..., 2]) and np.array_equal(res.obj_scores, rsc)


def test_multi_lane_equals_single_lane_on_changing_inputs(pkg):
    """The multi-stream schedule must give the single-stream result bit for bit on inputs and shapes that change
    from call to call (a missing cross-lane dependency shows up as a run-to-run difference, and repeated identical
    inputs would mask it)."""
    lib = pkg._lib.load()
    for C in (32, 48):
        net, _ = _net(pkg, C, 3)
        shapes = [(64, 128), (128, 256), (256, 512), (64, 64), (32, 32)]
        xs = {(s, r): torch.from_numpy(pkg.synth.synth_images(1, s[0], s[1], 5 + r)).to(DEV) for s in shapes for r in range(3)}
        lib.hh_set_multi_lane(net._h, 0)
        ref = {k: [t.clone() for t in net.forward_raw(x)] for k, x in xs.items()}
        lib.hh_set_multi_lane(net._h, 1)
        for rep in range(6):
            for k, x in xs.items():
                i, d = net.forward_raw(x)
                assert torch.equal(i, ref[k][0]) and torch.equal(d, ref[k][1]), (C, rep, k)


def test_evaluate_images_packs_what_single_calls_return(pkg):
    """bin/eval.py:18-49 through the drop-in wrapper: per-image results in COCO layout, in dataset order."""
    ev = importlib.import_module(PKG + ".keypoints.evaluation")
    net, _ = _net(pkg, 32, 4)
    model = pkg.InferenceKeypointsModel(net, det_thr=0.05, tag_thr=0.5, use_flip=False, input_size=128, device=DEV)
    rs = np.random.RandomState(3)
    images = [rs.randint(0, 255, (100 + 20 * i, 160, 3)).astype(np.uint8) for i in range(3)]
    out = ev.evaluate_images(model, images, [11, 12, 13])
    exp = []
    for i, im in enumerate(images):
        r = model(im, None)
        exp += ev.pack_coco_results(11 + i, r.kpts_coords, r.obj_scores)
    assert out == exp and len(out) >= 3 and len(out[0]["keypoints"]) == 51


@pytest.mark.parametrize("use_flip", [False, True])
def test_infer_images_batched_equals_single_calls(pkg, use_flip):
    """InferenceKeypointsModel.infer_images (shape buckets -> one forward + one decode per batch) returns for every image what
    the reference-style single call returns: same coordinates, scores, tags, person scores, bit for bit."""
    net, _ = _net(pkg, 32, 4)
    model = pkg.InferenceKeypointsModel(net, det_thr=0.05, tag_thr=0.5, use_flip=use_flip, input_size=128, device=DEV)
    rs = np.random.RandomState(7)
    shapes = [(96, 128), (128, 96), (96, 128), (100, 100), (96, 128), (128, 96), (96, 128)]
    images = [rs.randint(0, 255, s + (3,)).astype(np.uint8) for s in shapes]
    batched = model.infer_images(images, max_batch=3)  # the (96,128) bucket splits into batches of 3 + 1
    assert len(batched) == len(images)
    for im, rb in zip(images, batched):
        r1 = model(im, None)
        assert rb.raw_image is im and tuple(rb.model_input_image.shape) == tuple(r1.model_input_image.shape)
        for f in ("kpts_coords", "kpts_scores", "kpts_tags", "obj_scores"):
            a, b = getattr(rb, f), getattr(r1, f)
            assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), f
        assert torch.equal(rb.model_input_image, r1.model_input_image)


def _passthrough_net(pkg, C=32, seed=0):
    net = pkg.HigherHRNet(17, C)
    sd = {k: torch.from_numpy(v) for k, v in pkg.synth.synth_passthrough_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 17, seed).items()}
    net.load_state_dict(sd)
    return net.to(DEV).eval(), sd


def test_chained_forward_decode_on_person_like_maps(pkg):
    """forward -> decode as ONE chain (the decode consumes the engine's own bf16-path outputs through the channel-slice views
    HigherHRNet.forward returns) against fp32 oracle forward -> oracle decode, on a net whose outputs are person-like maps
    (synth pass-through weights: dense random channels beside reserved channels that carry constructed heatmaps / tags).
    What bf16 may change: the half-res heatmaps go through the transposed conv + BN + four residual units in bf16 (the
    quarter-res maps and tags are bf16-exact by construction), so peak positions / grouping must agree and sub-pixel
    offsets + scores must agree closely."""
    net, sd = _passthrough_net(pkg)
    B, hq, wq = 4, 64, 64
    imgs, hms, fields = pkg.synth.synth_passthrough_images(B, hq, wq, [3, 6, 10, 1], 17, 1)
    x = torch.from_numpy(imgs).to(DEV)
    (g_hq, g_hh), g_tags = net(x)
    assert np.array_equal(g_hq.cpu().numpy(), hms) and np.array_equal(g_tags.cpu().numpy(), np.repeat(fields[:, None], 17, 1))
    with torch.no_grad():
        (r_hq, r_hh), r_tags = ofw.higher_hrnet(torch.from_numpy(imgs), sd, 17)
    _close(g_hh.cpu().numpy(), r_hh.numpy(), "hm_h")
    parser = pkg.MPPEHeatmapParser(17, 30, 0.05, 0.5)
    got = parser.to_lists(*parser.decode_batch_device(g_hq, g_hh, [g_tags]))
    same_xy = total = 0
    max_dxy = max_ds = 0.0
    for b in range(B):
        rj, rs = orc.decode(r_hq[b].numpy(), r_hh[b].numpy(), [r_tags[b].numpy()], max_people=30, det_thr=0.05, tag_thr=0.5)
        j, s = got[b]
        assert j.shape == rj.shape, (b, j.shape, rj.shape)                   # same number of people
        assert np.array_equal(j[..., 2] > 0, rj[..., 2] > 0)                  # same joints present in the same groups
        same = (j[..., :2] == rj[..., :2]).all(-1)
        assert np.array_equal(j[..., 3:][same], rj[..., 3:][same]) and np.allclose(j[..., 3:], rj[..., 3:], atol=0.1)  # tags: exact at equal pixels
        max_dxy = max(max_dxy, float(np.abs(j[..., :2] - rj[..., :2]).max()))
        max_ds = max(max_ds, float(np.abs(j[..., 2] - rj[..., 2]).max()), float(np.abs(s - rs).max()))
        same_xy += int((j[..., :2] == rj[..., :2]).all(-1).sum()); total += j.shape[0] * j.shape[1]
    # a full-resolution peak is the x2 bilinear of a smooth half-res blob: its top is flat to within bf16's 0.4 %, so the
    # arg-max may move to a neighbouring pixel and the quarter-pixel offset may flip; grouping never changes
    print(f"chained: {same_xy}/{total} joint coordinates identical, max |dxy| {max_dxy}, max |dscore| {max_ds}")
    assert max_dxy <= 1.0 and max_ds <= 5e-3 and same_xy >= 0.93 * total, (same_xy, total, max_dxy, max_ds)  # measured: 719/748, 0.5 px, 7.6e-4
    # and the engine's decode equals the oracle's decode of the engine's own maps bit for bit (decode parity on real net outputs)
    for b in range(B):
        oj, os_ = orc.decode(g_hq[b].cpu().numpy(), g_hh[b].cpu().numpy(), [g_tags[b].cpu().numpy()], max_people=30, det_thr=0.05, tag_thr=0.5)
        assert np.array_equal(got[b][0], oj) and np.array_equal(got[b][1], os_)


def test_validation_step_decodes_at_the_validation_thresholds(pkg):
    """KeypointsModule.validation_step (keypoints/module.py:73-111): losses without a backward + one KeypointsResult per image
    decoded with max_num_people=20, det_thr=0.1, tag_thr=1.0; the batched decode equals per-image set_preds() and the oracle."""
    net, _ = _passthrough_net(pkg)
    kp = importlib.import_module(PKG + ".keypoints")
    res_mod = importlib.import_module(PKG + ".keypoints.results")
    B, S = 3, 128
    imgs = pkg.synth.synth_passthrough_images(B, S // 4, S // 4, [2, 4, 0], 17, 5)[0]
    hms, masks, joints = pkg.synth.synth_train_targets(B, 17, S, 3, seed=2)
    batch = (torch.from_numpy(imgs).to(DEV), [torch.from_numpy(h).to(DEV) for h in hms], [torch.from_numpy(m).to(DEV) for m in masks], joints)
    module = kp.KeypointsModule(kp.KeypointsModel(net), pkg.AEKeypointsLoss(), torch.optim.SGD(net.parameters(), lr=0.0))
    metrics, results = module.validation_step(batch)
    assert set(metrics) == {"loss", "hm_0_loss", "hm_1_loss", "push_0_loss", "pull_0_loss"} and all(np.isfinite(v) for v in metrics.values())
    assert abs(metrics["loss"] - (metrics["hm_0_loss"] + metrics["hm_1_loss"] + metrics["push_0_loss"] + metrics["pull_0_loss"])) < 1e-5 * max(1.0, metrics["loss"])
    assert len(results) == B and all(not p.requires_grad or p.grad is None for p in net.parameters())
    (g_hq, g_hh), g_tags = net(batch[0])
    for b, r in enumerate(results):
        assert isinstance(r, res_mod.KeypointsResult) and r.max_num_people == 20 and r.det_thr == 0.1 and r.tag_thr == 1.0
        oj, os_ = orc.decode(g_hq[b].cpu().numpy(), g_hh[b].cpu().numpy(), [g_tags[b].cpu().numpy()], max_people=20, det_thr=0.1, tag_thr=1.0)
        assert np.array_equal(r.kpts_coords, oj[..., :2]) and np.array_equal(r.kpts_scores, oj[..., 2]) and np.array_equal(r.obj_scores, os_)
        one = res_mod.KeypointsResult(r.model_input_image, r._kpts_heatmaps, r._tags_heatmaps, r.limbs, 20, 0.1, 1.0)
        one.set_preds()
        assert np.array_equal(one.kpts_coords, r.kpts_coords) and np.array_equal(one.kpts_tags, r.kpts_tags) and np.array_equal(one.obj_scores, r.obj_scores)


def _loss_case(pkg, case):
    tag, B, size, people, seed, holes = case
    hms, masks, joints = pkg.synth.synth_train_targets(B, 17, size, people, seed=seed, mask_holes=holes)
    joints = pkg.synth.edit_loss_case(tag, joints)
    pred, tags = pkg.synth.synth_train_preds(hms, seed)
    return hms, masks, joints, pred, tags


def test_ae_loss_and_gradients_match_reference(pkg):
    """AEKeypointsLoss.calculate_loss (loss.py:64-93) and the gradients of hm0+hm1+push+pull (module.py:50-59) through
    torch autograd on the HIP loss, against (i) the reference's values + autograd gradients (tests/golden/loss.npz) and
    (ii) the oracle on every element.  fp32; sums are taken in a different order, hence rtol 2e-5."""
    import json
    from oracle import loss as ol
    lossmod = importlib.import_module(PKG + ".keypoints.loss")
    g = np.load(os.path.join(GOLDEN, "loss.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "loss_meta.json")))
    fn = lossmod.AEKeypointsLoss()
    for case in meta["cases"]:
        tag = case[0]
        hms, masks, joints, pred, tags = _loss_case(pkg, case)
        K = 17
        # predictions laid out the way the net emits them: stage-0 heatmaps and tags are channel slices of one tensor
        init = torch.from_numpy(np.concatenate([pred[0], tags], 1)).to(DEV).requires_grad_()
        dec = torch.from_numpy(pred[1]).to(DEV).requires_grad_()
        hl, push, pull = fn.calculate_loss([init[:, :K], dec], init[:, K:], [torch.from_numpy(h).to(DEV) for h in hms],
                                           [torch.from_numpy(m).to(DEV) for m in masks], joints)
        total = hl[0] + hl[1] + push[0] + pull[0]
        total.backward()
        got = np.array([hl[0].item(), hl[1].item(), push[0].item(), pull[0].item(), total.item()], np.float32)
        np.testing.assert_allclose(got, g[f"{tag}.losses"], rtol=2e-5, atol=1e-9)
        ohl, opush, opull, ogp, ogt = ol.calculate_loss(pred, tags, hms, masks, joints)
        gi = init.grad.cpu().numpy()
        np.testing.assert_allclose(gi[:, :K], ogp[0], rtol=2e-5, atol=1e-10)
        np.testing.assert_allclose(dec.grad.cpu().numpy(), ogp[1], rtol=2e-5, atol=1e-10)
        np.testing.assert_allclose(gi[:, K:], ogt, rtol=2e-5, atol=1e-10)
        flat = gi[:, K:].ravel()
        assert np.array_equal(np.flatnonzero(flat), g[f"{tag}.g_tags_idx"])
        np.testing.assert_allclose(flat[g[f"{tag}.g_tags_idx"]], g[f"{tag}.g_tags_val"], rtol=2e-5, atol=1e-10)
        for i, gg in enumerate((gi[:, :K], dec.grad.cpu().numpy())):
            np.testing.assert_allclose(gg.ravel()[g[f"{tag}.g_pred{i}_idx"]], g[f"{tag}.g_pred{i}_val"], rtol=2e-5, atol=1e-10)
    # loss scaling arrives through grad_output (GradScaler), and error paths stay python exceptions
    t = torch.zeros(1, 17, 8, 8, device=DEV, requires_grad=True)
    pu, pl = lossmod.AEGroupingLoss()(t, [np.array([[[1, 1, 1]] * 17, [[2, 2, 1]] * 17], np.int32)])
    (pu * 1024.0).backward()
    assert abs(pu.item() - 0.5) < 1e-7 and pl.item() == 0 and float(t.grad.abs().max()) == 0.0  # equal tags: exp(0), zero slope
    # joints uploaded ahead of the step (DeviceJoints, e.g. for a captured step) give the same losses as the host list
    tt = torch.randn(2, 17, 8, 8, device=DEV)
    jl = [np.array([[[1, 1, 1]] * 17, [[5, 2, 1]] * 17], np.int32), np.array([[[3, 6, 1]] * 17], np.int32)]
    a = lossmod.AEGroupingLoss()(tt, jl)
    b = lossmod.AEGroupingLoss()(tt, lossmod.upload_joints(jl, 17, 8, 8, DEV))
    assert a[0].item() == b[0].item() and a[1].item() == b[1].item()
    with pytest.raises(IndexError):
        lossmod.AEGroupingLoss()(t, [np.array([[[8, 1, 1]] * 17], np.int32)])
    with pytest.raises(pkg._lib.HHError):
        lossmod.HeatmapsLoss()(torch.zeros(1, 17, 8, 8), torch.zeros(1, 17, 8, 8), torch.ones(1, 8, 8))


def test_full_size_properties_batch32_512(pkg):
    """BASELINE.json configs[1] at its full size (B=32, 512x512, W32), where the CPU oracle would take minutes: size-independent
    properties instead.  (i) images are independent: row b of a batched forward is bit-identical to the forward of image b alone,
    and repeated runs are bit-identical; (ii) decode of a batch equals decode of each image alone; (iii) top-k scores come out
    sorted, are local 5x5 maxima of the aggregated map and no joint of a decoded person lies outside the image."""
    net, _ = _net(pkg, 32, 0)
    B, H, W, K = 32, 512, 512, 17
    x = torch.from_numpy(pkg.synth.synth_images(B, H, W, seed=3)).to(DEV)
    init, dec = net.forward_raw(x)
    init2, dec2 = net.forward_raw(x)
    assert torch.equal(init, init2) and torch.equal(dec, dec2)
    assert torch.isfinite(init).all() and torch.isfinite(dec).all()
    for b in (0, 13, 31):
        i1, d1 = net.forward_raw(x[b:b + 1].contiguous())
        assert torch.equal(i1[0], init[b]) and torch.equal(d1[0], dec[b])
    uniq = [pkg.synth.synth_decode_maps(K, H // 4, W // 4, 10, seed=2000 + i)[:3] for i in range(4)]
    hm_q = torch.from_numpy(np.stack([uniq[i % 4][0] for i in range(B)])).to(DEV)
    hm_h = torch.from_numpy(np.stack([uniq[i % 4][1] for i in range(B)])).to(DEV)
    tags = torch.from_numpy(np.stack([uniq[i % 4][2][0] for i in range(B)])).to(DEV)
    parser = pkg.MPPEHeatmapParser(K, 30, 0.05, 0.5)
    joints, scores, num, flags = [t.clone() for t in parser.decode_batch_device(hm_q, hm_h, [tags])]
    assert not flags.any()
    parser.set_exact_topk(True)   # every tile processed: identical results, and the full candidate lists are available
    exact = parser.decode_batch_device(hm_q, hm_h, [tags])
    assert all(torch.equal(a, b) for a, b in zip((joints, scores, num, flags), exact))
    _, _, sk = parser.last_top_k(B, 1)
    parser.set_exact_topk(False)
    assert np.all(np.diff(sk, axis=-1) <= 0)  # per joint: candidates in descending score order
    n = num.cpu().numpy()
    assert n.min() >= 1 and n.max() <= 30
    j = joints.cpu().numpy()
    for b in range(B):
        p = j[b, : n[b]]
        seen = p[..., 2] > 0
        assert np.all(p[..., 0][seen] >= 0) and np.all(p[..., 0][seen] < W) and np.all(p[..., 1][seen] >= 0) and np.all(p[..., 1][seen] < H)
        assert np.array_equal(j[b], j[b % 4]) and n[b] == n[b % 4]  # the batch repeats 4 distinct images
    for b in (0, 5):
        j1, s1, n1, _ = parser.decode_batch_device(hm_q[b:b + 1], hm_h[b:b + 1], [tags[b:b + 1]])
        assert int(n1[0]) == int(n[b]) and torch.equal(j1[0], joints[b]) and torch.equal(s1[0], scores[b])
    # the oracle agrees on one of the full-size images (takes ~0.3 s per image)
    rj, rs = orc.decode(uniq[0][0], uniq[0][1], [uniq[0][2][0]], max_people=30, det_thr=0.05, tag_thr=0.5)
    assert np.array_equal(j[0, : n[0]], rj) and np.array_equal(scores[0, : n[0]].cpu().numpy(), rs)


def _bf(t):  # round to bf16 and back: the values the kernels actually see
    return t.to(torch.bfloat16).float()


def test_training_building_blocks_conv_and_batchnorm(pkg):
    """hh_conv2d (forward with the current fp32 weights, stride 1/2, 1x1/3x3, and the stride-1 data gradient) and train-mode
    BatchNorm forward/backward against torch fp32 on the same bf16-rounded operands.  Tolerance: bf16 output rounding
    (2^-8 relative) plus fp32 accumulation order -> 1.5 % of the tensor's max."""
    ops = importlib.import_module(PKG + ".keypoints.train_ops")
    F = torch.nn.functional
    g = torch.Generator().manual_seed(0)

    def close(got, ref, what, tol=1.5e-2, min_cos=0.9995):
        """max error relative to the tensor's max AND direction: cosine with the fp32 reference (a wrong term in a backward
        formula that stays inside a max-error band still turns the vector: measured cosines are >= 0.99998)."""
        got, ref = got.float().cpu(), ref.float()
        err = (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-6)
        cos = float(torch.dot(got.flatten().double(), ref.flatten().double()) / (got.double().norm() * ref.double().norm() + 1e-30))
        assert err < tol and cos > min_cos, (what, err, cos)

    packed_cases = []
    for (cin, cout, ks, stride, hw) in [(64, 64, 3, 1, 32), (32, 32, 3, 1, 40), (128, 64, 1, 1, 16), (64, 256, 1, 1, 24), (64, 128, 3, 2, 32), (32, 32, 3, 2, 24),
                                        (48, 48, 3, 1, 16), (256, 256, 3, 1, 16)]:
        x = _bf(torch.randn(2, cin, hw, hw, generator=g))
        w = _bf(torch.randn(cout, cin, ks, ks, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5)
        b = torch.randn(cout, generator=g)
        res = _bf(torch.randn(2, cout, hw // stride, hw // stride, generator=g))
        ref = F.relu(F.conv2d(x, w, b, stride, (ks - 1) // 2) + res)
        got = ops.conv2d(x.to(DEV, torch.bfloat16), w.to(DEV), stride, bias=b.to(DEV), res=res.to(DEV, torch.bfloat16), relu=True)
        assert got.shape == ref.shape
        close(got, ref, ("conv", cin, cout, ks, stride))
        # gradients = torch autograd of the same conv on the same bf16-rounded operands
        xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
        dy = _bf(torch.randn(2, cout, hw // stride, hw // stride, generator=g))
        F.conv2d(xr, wr, None, stride, (ks - 1) // 2).backward(dy)
        close(ops.conv2d_weight_grad(x.to(DEV, torch.bfloat16), dy.to(DEV, torch.bfloat16), ks, stride), wr.grad, ("wgrad", cin, cout, ks, stride), 5e-3)
        dgrad = ops.conv2d(dy.to(DEV, torch.bfloat16), w.to(DEV), stride, data_grad=True)
        close(dgrad, xr.grad, ("dgrad", cin, cout, ks, stride))
        packed_cases.append((w.to(DEV), stride, x.to(DEV, torch.bfloat16), dy.to(DEV, torch.bfloat16),
                             ops.conv2d(x.to(DEV, torch.bfloat16), w.to(DEV), stride), dgrad))
    # weights packed ahead by ONE batched launch (hh_pack_conv_weights_batch + hh_conv2d_packed): bit-identical results
    pw = ops.PackedConvWeights([e for w, stride, *_ in packed_cases for e in ((w, stride, False), (w, stride, True))])
    pw.refresh()
    for i, (w, stride, xd, dyd, fwd, dgrad) in enumerate(packed_cases):
        assert torch.equal(ops.conv2d(xd, w, stride, packed=pw.buffers[2 * i]), fwd), ("packed fwd", tuple(w.shape), stride)
        assert torch.equal(ops.conv2d(dyd, w, stride, data_grad=True, packed=pw.buffers[2 * i + 1]), dgrad), ("packed dgrad", tuple(w.shape), stride)
    for (C, hw, relu, with_res) in [(32, 24, True, False), (64, 16, True, True), (256, 8, False, False), (48, 12, True, True), (384, 8, True, False)]:
        x = _bf(torch.randn(3, C, hw, hw, generator=g) * 2 + 0.5).requires_grad_()
        gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
        gamma.requires_grad_(); beta.requires_grad_()
        res = _bf(torch.randn(3, C, hw, hw, generator=g)).requires_grad_() if with_res else None
        z = F.batch_norm(x, None, None, gamma, beta, True, 0.0, 1e-5)
        ref = z + res if with_res else z
        ref = F.relu(ref) if relu else ref
        y, mean, invstd = ops.bn_train_forward(x.detach().to(DEV, torch.bfloat16), gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5,
                                               res.detach().to(DEV, torch.bfloat16) if with_res else None, relu)
        close(y, ref.detach(), ("bn fwd", C))
        np.testing.assert_allclose(mean.cpu().numpy(), x.detach().mean((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(invstd.cpu().numpy(), (x.detach().var((0, 2, 3), unbiased=False) + 1e-5).rsqrt().numpy(), rtol=1e-4)
        dy = _bf(torch.randn(3, C, hw, hw, generator=g))
        # reference backward from the SAME rounded forward output the kernel saw (its ReLU mask is y > 0 on the bf16 y)
        ref.backward(dy)
        dx, dgamma, dbeta, dres = ops.bn_train_backward(x.detach().to(DEV, torch.bfloat16), y, dy.to(DEV, torch.bfloat16), mean, invstd,
                                                        gamma.detach().to(DEV), relu, want_dres=with_res)
        close(dx, x.grad, ("bn dx", C), 2e-2)
        close(dgamma, gamma.grad, ("bn dgamma", C), 2e-2)
        close(dbeta, beta.grad, ("bn dbeta", C), 2e-2)
        if with_res:
            close(dres, res.grad, ("bn dres", C), 2e-2)
        else:  # without a residual the backward can do without y (mask recomputed from x by the forward's own function): the same bits
            px, pg, pb, _ = ops.bn_train_backward(x.detach().to(DEV, torch.bfloat16), None, dy.to(DEV, torch.bfloat16), mean, invstd,
                                                  gamma.detach().to(DEV), relu, beta=beta.detach().to(DEV))
            assert torch.equal(px, dx) and torch.equal(pg, dgamma) and torch.equal(pb, dbeta), ("bn backward without y", C, relu)


def test_fusion_sum_training_op_matches_torch(pkg):
    """hh_fusion_sum_forward / _backward (FusionLayer's relu(sum of terms) with nn.Upsample(nearest) folded into the read,
    hrnet.py:200-229) against torch fp32 autograd on the same bf16-rounded operands."""
    ops = importlib.import_module(PKG + ".keypoints.train_ops")
    F = torch.nn.functional
    g = torch.Generator().manual_seed(1)
    for C, H, W, shifts in [(32, 16, 24, [0, 1, 2, 3]), (64, 8, 8, [0, 0, 1]), (128, 4, 8, [0, 0, 0, 1]), (48, 8, 8, [0, 2])]:
        terms = [_bf(torch.randn(2, C, H >> s, W >> s, generator=g)).requires_grad_() for s in shifts]
        ref = F.relu(sum(t if s == 0 else F.interpolate(t, scale_factor=2 ** s, mode="nearest") for t, s in zip(terms, shifts)))
        dev = [t.detach().to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last) for t in terms]
        out = ops.fusion_sum(dev, shifts, relu=True)
        assert (out.float().cpu() - ref.detach()).abs().max().item() <= 1.6e-2 * ref.abs().max().item()  # one bf16 rounding of the fp32 sum
        dy = _bf(torch.randn(2, C, H, W, generator=g))
        # reference backward with the mask of the kernel's own (bf16) output, as the BatchNorm test does
        (ref * 0 + sum(t if s == 0 else F.interpolate(t, scale_factor=2 ** s, mode="nearest") for t, s in zip(terms, shifts))
         ).backward(dy * (out.float().cpu() > 0))
        grads = ops.fusion_sum_backward(dy.to(DEV, torch.bfloat16).contiguous(memory_format=torch.channels_last), out, shifts, relu=True)
        for t, gr, s in zip(terms, grads, shifts):
            a, b = gr.float().cpu(), t.grad
            assert a.shape == b.shape and (a - b).abs().max().item() <= 8e-3 * max(b.abs().max().item(), 1e-6), (C, s)


def test_running_statistics_match_torch_batchnorm(pkg):
    """The training forward's BatchNorm bookkeeping (train_net.bn + flush_running_stats: the unbiased variance is rebuilt from
    the kernel's invstd as (1 / invstd^2 - eps) * n / (n - 1)) against nn.BatchNorm2d on the same bf16-rounded input, over
    three steps with changing inputs: running_mean, running_var, num_batches_tracked."""
    from torch import nn
    tn = importlib.import_module(PKG + ".keypoints.train_net")
    g = torch.Generator().manual_seed(3)
    for C, hw, B in [(32, 24, 4), (64, 8, 2), (256, 4, 2)]:
        ours, ref = nn.BatchNorm2d(C).to(DEV), nn.BatchNorm2d(C)
        with torch.no_grad():
            for m in (ours, ref):
                m.weight.copy_(torch.linspace(0.5, 1.5, C)); m.bias.copy_(torch.linspace(-0.2, 0.2, C))
        ours.train(); ref.train()
        for step in range(3):
            x = _bf(torch.randn(B, C, hw, hw, generator=g) * (1 + step) + 0.3 * step)
            y = tn.bn(x.to(DEV, torch.bfloat16), ours, relu=False)
            tn.flush_running_stats()
            yr = ref(x)
            assert (y.float().cpu() - yr).abs().max().item() < 1.5e-2 * yr.abs().max().item()
        np.testing.assert_allclose(ours.running_mean.cpu().numpy(), ref.running_mean.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ours.running_var.cpu().numpy(), ref.running_var.numpy(), rtol=2e-4, atol=1e-6)
        assert int(ours.num_batches_tracked) == int(ref.num_batches_tracked) == 3


def test_train_step_batch8_matches_reference_autograd_tightly(pkg):
    """The same end-to-end comparison at batch 8 (tests/golden/train_step_b8.npz: the reference net in .train() mode + torch
    autograd): with 8 images the lowest-resolution branch normalises over 128 samples per channel instead of 32, the batch
    statistics stop moving with bf16 rounding, and the tolerance is what bf16 activations through ~110 conv + BN layers leave:
    loss within 0.3 %, outputs within 4 % of max, the gradient norm of every one of the 907 parameters within 10 % (median within
    2 %) and its direction against the fp32 oracle's full gradient at cosine > 0.92 (median > 0.98)."""
    g = np.load(os.path.join(GOLDEN, "train_step_b8.npz"))
    K = 17
    net = pkg.HigherHRNet(K, 32)
    sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 5)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = torch.from_numpy(pkg.synth.synth_images(8, 128, 128, seed=1))
    hms, tags = net(x.to(DEV))
    loss = (hms[0] ** 2).mean() + (hms[1] ** 2).mean() + (tags ** 2).mean()
    loss.backward()
    dl = abs(loss.item() - float(g["loss"])) / float(g["loss"])
    eo = []
    for name, t in (("hm0", hms[0]), ("hm1", hms[1]), ("tags", tags)):
        a = t.detach().float().cpu().numpy().ravel()
        eo.append(float(np.abs(a[g[f"{name}.idx"]] - g[f"{name}.val"]).max() / float(g[f"{name}.absmax"])))
    names = [str(n) for n in g["grad.names"]]
    params = dict(net.named_parameters())
    ratios = np.array([params[n].grad.double().norm().item() / max(g["grad.norms"][i], 1e-30) for i, n in enumerate(names)])
    osd = {k: (v.clone().float().requires_grad_() if k in params else v.clone()) for k, v in sd.items()}
    oh, ot = ofw.higher_hrnet(x, osd, K, train=True)
    ((oh[0] ** 2).mean() + (oh[1] ** 2).mean() + (ot ** 2).mean()).backward()
    cos = np.array([float(torch.dot(params[n].grad.float().cpu().flatten(), osd[n].grad.flatten()) /
                          (params[n].grad.float().cpu().norm() * osd[n].grad.norm() + 1e-30)) for n in names])
    print(f"train step B=8: loss rel {dl:.4f}, outputs max/absmax {eo}, grad norm ratio [{ratios.min():.3f}, {ratios.max():.3f}] median {np.median(ratios):.3f}, "
          f"cosine min {cos.min():.4f} (param {names[int(cos.argmin())]}) median {np.median(cos):.4f}")
    assert dl < 3e-3 and max(eo) < 4e-2, (dl, eo)
    assert np.all((ratios > 0.90) & (ratios < 1.10)) and abs(np.median(ratios) - 1) < 0.02, (ratios.min(), ratios.max(), np.median(ratios))
    # measured: min 0.936 (the stem's and stage 0's weights: their gradients have crossed all ~110 layers backwards in bf16), median 0.987;
    # at batch 2 the same quantities are 0.85 / 0.95 -- the difference is the batch statistics, not the kernels
    assert cos.min() > 0.92 and np.median(cos) > 0.98, (cos.min(), np.median(cos))


def test_train_step_matches_reference_autograd(pkg):
    """HigherHRNet in .train() mode on the HIP training kernels (bf16 activations, batch-statistics BatchNorm, fp32 parameter
    gradients) against (i) the reference net in .train() mode + torch autograd (tests/golden/train_step.npz) and (ii) the
    fp32 oracle's full gradients.  bf16 through ~110 conv+BN layers with batch statistics over 2 images: outputs within
    8 % of max, loss within 0.5 %, gradient direction cosine > 0.85 for every parameter (median > 0.95) and norm ratios
    within 25 %; then one Adam step and an eval-mode forward on the updated weights."""
    g = np.load(os.path.join(GOLDEN, "train_step.npz"))
    K = 17
    net = pkg.HigherHRNet(K, 32)
    sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 5)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net = net.to(DEV).train()
    x = torch.from_numpy(pkg.synth.synth_images(2, 128, 128, seed=1))
    hms, tags = net(x.to(DEV))
    assert hms[0].shape == (2, K, 32, 32) and hms[1].shape == (2, K, 64, 64) and tags.shape == (2, K, 32, 32)
    loss = (hms[0] ** 2).mean() + (hms[1] ** 2).mean() + (tags ** 2).mean()
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 5e-3 * float(g["loss"])
    for name, t in (("hm0", hms[0]), ("hm1", hms[1]), ("tags", tags)):
        a = t.detach().float().cpu().numpy().ravel()
        assert np.abs(a[g[f"{name}.idx"]] - g[f"{name}.val"]).max() < 8e-2 * float(g[f"{name}.absmax"]), name
    # Beside the reference's OWN mixed-precision step (tests/golden/train_step_autocast.npz: the reference net under
    # torch.autocast(float16) + loss scaling, keypoints/module.py:48-60, run on CPU autocast): that step deviates from fp32 by
    # 0.3-0.4 % rms on the outputs (fp16 keeps 11 significand bits), this one keeps 8 (bf16) and no loss scale: its deviation
    # is larger by those three bits: 8x expected, 7.1-7.4x measured, bounded at 12x.
    ga = np.load(os.path.join(GOLDEN, "train_step_autocast.npz"))
    assert abs(loss.item() - float(ga["loss"])) < 5e-3 * float(ga["loss"])
    ratios_out = []
    for name, t in (("hm0", hms[0]), ("hm1", hms[1]), ("tags", tags)):
        a = t.detach().float().cpu().numpy().ravel()[g[f"{name}.idx"]]
        e_eng = np.sqrt(((a - g[f"{name}.val"]) ** 2).mean())
        e_ref16 = np.sqrt(((ga[f"{name}.val"] - g[f"{name}.val"]) ** 2).mean())
        ratios_out.append(e_eng / e_ref16)
    print("train step: rms deviation from fp32, engine bf16 / reference fp16-autocast:", [round(float(r), 1) for r in ratios_out])
    assert max(ratios_out) < 12.0, ratios_out  # measured 7.1-7.4: the three significand bits between fp16 and bf16
    names = [str(n) for n in g["grad.names"]]
    params = dict(net.named_parameters())
    assert set(names) == set(params) and all(p.grad is not None for p in params.values())
    ratios = np.array([params[n].grad.double().norm().item() / max(g["grad.norms"][i], 1e-30) for i, n in enumerate(names)])
    assert np.all((ratios > 0.75) & (ratios < 1.33)) and abs(np.median(ratios) - 1) < 0.05, (ratios.min(), ratios.max())
    # full gradients of the fp32 oracle (autograd on the CPU)
    osd = {k: (v.clone().float().requires_grad_() if k in params else v.clone()) for k, v in sd.items()}
    oh, ot = ofw.higher_hrnet(x, osd, K, train=True)
    ((oh[0] ** 2).mean() + (oh[1] ** 2).mean() + (ot ** 2).mean()).backward()
    cos = []
    for n in names:
        a, b = params[n].grad.float().cpu().flatten(), osd[n].grad.flatten()
        cos.append(float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30)))
    assert min(cos) > 0.85 and np.median(cos) > 0.95, (min(cos), np.median(cos))
    # BatchNorm running statistics moved the way nn.BatchNorm2d moves them (momentum 0.1, unbiased variance)
    st = net.state_dict()
    for k in ("backbone.bn1.running_mean", "backbone.bn1.running_var", "deconv_layers.0.deconv.1.running_mean",
              "backbone.stages.3.blocks.4.scales_blocks.3.3.bn2.running_var"):
        np.testing.assert_allclose(st[k].cpu().numpy(), g["stat." + k], rtol=3e-2, atol=3e-3)
    assert int(st["backbone.bn1.num_batches_tracked"]) == 1
    # the parameters are ordinary fp32 nn.Parameters: a torch optimizer steps them, and eval mode re-folds the new weights
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    before = params["backbone.conv1.weight"].detach().clone()
    opt.step()
    assert not torch.equal(before, params["backbone.conv1.weight"])
    net.eval()
    with torch.no_grad():
        eh, et = net(x.to(DEV))
    assert torch.isfinite(eh[0]).all() and torch.isfinite(eh[1]).all() and eh[0].shape == hms[0].shape


def test_train_step_under_distributed_data_parallel(pkg):
    """base/model.py:36-48 wraps the net in DistributedDataParallel: the HIP training forward must survive the wrapper
    (parameter registration, autograd hooks firing on the custom Functions' parameter gradients, bucketed all-reduce on
    RCCL).  One rank here (one GPU per box): the reduced gradients must equal the plain module's."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    K = 17
    def make():
        net = pkg.HigherHRNet(K, 32)
        net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 5)) for k, v in net.state_dict().items()})
        return net.to(DEV).train()
    x = torch.from_numpy(pkg.synth.synth_images(2, 64, 64, seed=2)).to(DEV)
    def grads(m):
        hms, tags = m(x)
        ((hms[0] ** 2).mean() + (hms[1] ** 2).mean() + (tags ** 2).mean()).backward()
    plain = make()
    grads(plain)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29533", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        ddp = DDP(make(), device_ids=[0])
        grads(ddp)
        torch.cuda.synchronize()
        for (n, a), (_, b) in zip(plain.named_parameters(), ddp.module.named_parameters()):
            assert b.grad is not None and torch.equal(a.grad, b.grad), n
        # KeypointsModel.to_DDP with bf16 gradient buckets: the same gradients to bf16 precision
        model = importlib.import_module(PKG + ".keypoints.model").KeypointsModel(make())
        model.to_DDP(0, use_batchnorm=False, bf16_gradients=True)
        grads(model.net)
        torch.cuda.synchronize()
        for (n, a), (_, b) in zip(plain.named_parameters(), model.net.module.named_parameters()):
            assert b.grad is not None and (a.grad - b.grad).abs().max() <= 8e-3 * a.grad.abs().max() + 1e-12, n
    finally:
        if created:
            dist.destroy_process_group()


def test_sync_batchnorm_two_ranks_equal_the_full_batch(pkg):
    """`to_DDP(device_id, use_batchnorm=True)` (base/model.py:36-48, the reference trainer's default) shares the BatchNorm
    statistics across ranks.  Two ranks (both on this GPU, gloo) each take half a batch: op outputs / gradients, DDP-averaged
    parameter gradients and running statistics must equal the single-process full-batch ones (tests/syncbn_worker.py)."""
    import subprocess, sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "syncbn_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", "29541"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    print("\n".join(outs))
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_sync_batchnorm_entry_points_without_exchange_equal_the_fused_pass(pkg):
    """With count == P and no all-reduce the split entry points are the fused BatchNorm passes, bit for bit."""
    ops = importlib.import_module("pytorch-human-pose_amd.keypoints.train_ops")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 48, 20, 28, generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
    r = torch.randn(2, 48, 20, 28, generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(2, 48, 20, 28, generator=g).to(torch.bfloat16).to(DEV).contiguous(memory_format=torch.channels_last)
    gamma, beta = (torch.rand(48, generator=g) + 0.5).to(DEV), torch.randn(48, generator=g).to(DEV)
    saved = ops._all_reduce_sums
    ops._all_reduce_sums = lambda sums, group: None
    try:
        y1, m1, i1, cnt = ops.sync_bn_train_forward(x, gamma, beta, 1e-5, r, True, None, 1)
        b1 = ops.sync_bn_train_backward(x, y1, dy, m1, i1, gamma, True, True, None, cnt)
    finally:
        ops._all_reduce_sums = saved
    y0, m0, i0 = ops.bn_train_forward(x, gamma, beta, 1e-5, r, True)
    b0 = ops.bn_train_backward(x, y0, dy, m0, i0, gamma, True, want_dres=True)
    assert torch.equal(y0, y1) and torch.equal(m0, m1) and torch.equal(i0, i1)
    for a, b in zip(b0, b1):
        assert torch.equal(a, b)


@pytest.mark.parametrize("C,K", [(48, 17), (32, 5)])
def test_train_forward_backward_other_widths_and_shapes(pkg, C, K):
    """Channel widths that are not multiples of 32 (W48), another joint count and a non-square input go through the training
    kernels (zero-padded channel counts, 2x2 phase convs of the transposed conv): every parameter gets a finite gradient."""
    net = pkg.HigherHRNet(K, C)
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 1)) for k, v in net.state_dict().items()})
    net = net.to(DEV).train()
    x = torch.from_numpy(pkg.synth.synth_images(2, 128, 192, 3)).to(DEV)
    hms, tags = net(x)
    assert hms[0].shape == (2, K, 32, 48) and hms[1].shape == (2, K, 64, 96) and tags.shape == (2, K, 32, 48)
    ((hms[0] ** 2).mean() + (hms[1] ** 2).mean() + (tags ** 2).mean()).backward()
    for n, p in net.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, n


def test_training_loop_reduces_the_loss(pkg):
    """module.py:43-71 end to end on the HIP path: forward (batch-stat BN) + AEKeypointsLoss + backward + Adam on one fixed
    synthetic batch; the loss must fall by more than 5x in 15 steps (it goes 4.3 -> 0.4), i.e. the gradients are useful,
    not merely close to the reference at step 0."""
    K, S, B = 17, 128, 4
    net = pkg.HigherHRNet(K, 32)
    net.load_state_dict({k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()})
    # through the reference's own wrappers: KeypointsModel (model.py:15-40) + KeypointsModule.training_step (module.py:43-71)
    km = importlib.import_module(PKG + ".keypoints.model")
    model = km.KeypointsModel(net)
    model.to_CUDA(0)
    model.net.train()
    module = km.KeypointsModule(model, pkg.AEKeypointsLoss(), torch.optim.Adam(model.net.parameters(), lr=1e-3))
    x = torch.from_numpy(pkg.synth.synth_images(B, S, S, 0))
    hms, masks, joints = pkg.synth.synth_train_targets(B, K, S, 3, seed=0)
    batch = module.batch_to_device((x, [torch.from_numpy(h) for h in hms], [torch.from_numpy(m) for m in masks], joints))
    assert batch[0].device.type == "cuda" and batch[1][0].device.type == "cuda"
    metrics = [module.training_step(batch, i) for i in range(15)]
    losses = [m["loss"] for m in metrics]
    assert set(metrics[0]) == {"loss", "hm_0_loss", "hm_1_loss", "push_0_loss", "pull_0_loss"}
    assert abs(metrics[0]["loss"] - sum(v for k, v in metrics[0].items() if k != "loss")) < 1e-4 * abs(metrics[0]["loss"]) + 1e-6
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] / 5, losses


# ---------------------------------------------------------------------------------------------------------------------
# fp8 conv path (BASELINE.json configs[4]).  No reference precedent (the reference infers in fp32): parity is a STATED
# tolerance against the fp32 goldens / oracle.  e4m3 keeps 3 mantissa bits: rounding a tensor to it costs 3.6 % rms, so a conv
# whose weights and inputs are e4m3 leaves ~5 % on its output.  Round 3 keeps the residual trunk (block outputs, fusion sums) in
# bf16 beside the e4m3 copy the next conv reads, and runs the three heads on the bf16 kernels: every block adds the error of its
# two convs to the trunk instead of also re-rounding the trunk itself.  tools/probes/fp8_emulate.py reproduces both plans on the
# CPU oracle with fake quantisation: 16-18 % rms at the outputs for the round-2 plan (what the round-2 engine measured), 6-9 %
# for this one, of which ~5 % is the e4m3 WEIGHTS alone (activations in fp32) -- the floor of the format on these seeded nets.
# Stated tolerance: rms error <= 0.10 * rms(reference), max error <= 0.20 * max |reference|, correlation >= 0.99.
FP8_TOL_MAX, FP8_TOL_RMS, FP8_MIN_CORR = 0.20, 0.10, 0.99


def _fp8_net(pkg, C, seed, calib_shape):
    net = pkg.HigherHRNet(17, C, dtype="fp8")
    sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, seed)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    net.calibrate(torch.from_numpy(pkg.synth.synth_images(*calib_shape, seed=4242)).to(DEV))
    return net, sd


def _fp8_close(got, ref, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, what
    emax = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)
    erms = np.sqrt(((got - ref) ** 2).mean()) / max(np.sqrt((ref**2).mean()), 1e-6)
    corr = np.corrcoef(got.reshape(-1), ref.reshape(-1))[0, 1]
    assert emax <= FP8_TOL_MAX and erms <= FP8_TOL_RMS and corr >= FP8_MIN_CORR, f"{what}: max {emax:.3f} rms {erms:.3f} corr {corr:.4f}"
    return emax, erms, corr


@pytest.mark.parametrize("tag,C,B,H,W,seed", [("w48_64", 48, 1, 64, 64, 3), ("w32_128", 32, 2, 128, 128, 1), ("w32_96x160", 32, 1, 96, 160, 2)])
def test_fp8_forward_vs_reference_golden(pkg, net_golden, tag, C, B, H, W, seed):
    net, _ = _fp8_net(pkg, C, seed, (4, H, W))
    x = torch.from_numpy(pkg.synth.synth_images(B, H, W, seed)).to(DEV)
    for use_graph in (False, True, True):  # eager, capture, replay
        net.use_graph = use_graph
        hms, tags = net(x)
        _fp8_close(hms[0].cpu().numpy(), net_golden[f"{tag}/hm_q"], "hm_q")
        _fp8_close(hms[1].cpu().numpy(), net_golden[f"{tag}/hm_h"], "hm_h")
        _fp8_close(tags.cpu().numpy(), net_golden[f"{tag}/tags"], "tags")


def test_fp8_w48_640_vs_oracle_and_batch_consistency(pkg):
    """The configuration BASELINE.json names (W48 @ 640x640) at batch 3 against the fp32 oracle on the same weights; images are
    independent, so the same image in two batch slots gives identical bits; multi-lane = single-lane bit for bit."""
    net, sd = _fp8_net(pkg, 48, 5, (2, 640, 640))
    x = torch.from_numpy(pkg.synth.synth_images(3, 640, 640, 9)).to(DEV)
    x[2] = x[0]
    hms, tags = net(x)
    with torch.no_grad():
        rh, rt = ofw.higher_hrnet(x[:1].cpu(), sd, 17)
    _fp8_close(hms[0][:1].cpu().numpy(), rh[0].numpy(), "hm_q")
    _fp8_close(hms[1][:1].cpu().numpy(), rh[1].numpy(), "hm_h")
    _fp8_close(tags[:1].cpu().numpy(), rt.numpy(), "tags")
    assert torch.equal(hms[0][0], hms[0][2]) and torch.equal(hms[1][0], hms[1][2]) and torch.equal(tags[0], tags[2])
    a = [t.clone() for t in net.forward_raw(x)]
    pkg._lib.load().hh_set_multi_lane(net._h, 0)
    b = net.forward_raw(x)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_fp8_bf16_trunk_beats_the_all_e4m3_plan(pkg, net_golden):
    """The A/B behind the plan: HH_FP8_TRUNK=e4m3 HH_FP8_HEADS=e4m3 is round 2's plan (every tensor e4m3).  On the same weights,
    calibration batch and input the default plan's output error must be well below it (emulation: 0.4-0.5x)."""
    x = torch.from_numpy(pkg.synth.synth_images(2, 128, 128, 1)).to(DEV)
    errs = {}
    for name, env in (("trunk16", {}), ("all8", {"HH_FP8_TRUNK": "e4m3", "HH_FP8_HEADS": "e4m3"})):
        with _switch_env(env):
            net, _ = _fp8_net(pkg, 32, 1, (4, 128, 128))
        hms, tags = net(x)
        e = []
        for got, ref in ((hms[0], net_golden["w32_128/hm_q"]), (hms[1], net_golden["w32_128/hm_h"]), (tags, net_golden["w32_128/tags"])):
            got = got.cpu().numpy().astype(np.float64)
            e.append(np.sqrt(((got - ref) ** 2).mean()) / np.sqrt((ref.astype(np.float64) ** 2).mean()))
        errs[name] = e
    print("fp8 rms errors (hm_q, hm_h, tags):", errs)
    assert all(a <= FP8_TOL_RMS for a in errs["trunk16"]), errs
    assert all(a < 0.7 * b for a, b in zip(errs["trunk16"], errs["all8"])), errs


def test_fp8_chained_forward_decode(pkg):
    """fp8 forward -> hh_decode as one chain on the pass-through net (constructed people carried from the input images to the
    output maps).  What can be asserted, and what cannot:
      * the carried heatmap / tag values arrive within the e4m3 steps of the tensors they crossed (the stem's two convs write
        e4m3; relative step 2^-3, i.e. <= 6.25 % per rounding) -- the fp8 convs beside them leak nothing
        into the reserved channels and every scale is applied and undone correctly;
      * hh_decode of the fp8 maps equals the oracle's decode of those same maps bit for bit;
      * no person is lost: every person the fp32 maps decode to has an fp8 counterpart holding >= 80 % of its detected joints within
        2 px (measured: 94 %);
      * NOT the fp32 group COUNT: a value CARRIED through an e4m3 tensor is rounded to 3 mantissa bits, which turns the smooth top of
        a blob into a plateau of equal values, and every plateau pixel survives `maxpool == hm` as a peak (tools/probes/
        fp8_emulate.py-style check on the CPU oracle: rounding only the carried inputs to e4m3 already turns 6 / 4 / 1 / 4 groups
        into 10 / 8 / 2 / 8).  That is an artefact of carrying values through activations; a trained net's heatmaps come out of the
        fp32 accumulators of the last conv (measured output error 5-8 % rms, above).  Whether AE grouping at tag_thr 0.5 survives
        that error needs a trained checkpoint, which is not available offline: the fp8 path stays labelled experimental."""
    net = pkg.HigherHRNet(17, 32, dtype="fp8")
    sd = {k: torch.from_numpy(v) for k, v in pkg.synth.synth_passthrough_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, 17, 0).items()}
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    B, hq, wq = 4, 64, 64
    imgs, hms, fields = pkg.synth.synth_passthrough_images(B, hq, wq, [3, 2, 1, 3], 17, 1)
    cal = pkg.synth.synth_passthrough_images(4, hq, wq, [3, 2, 1, 3], 17, 9)[0]
    net.calibrate(torch.from_numpy(np.concatenate([cal, imgs])).to(DEV))
    (g_hq, g_hh), g_tags = net(torch.from_numpy(imgs).to(DEV))
    with torch.no_grad():
        (r_hq, r_hh), r_tags = ofw.higher_hrnet(torch.from_numpy(imgs), sd, 17)
    for got, ref, name in ((g_hq, r_hq, "hm_q"), (g_tags, r_tags, "tags"), (g_hh, r_hh, "hm_h")):
        got, ref = got.cpu().numpy().astype(np.float64), ref.numpy().astype(np.float64)
        # one e4m3 step of the value (+ the subnormal floor of the tensor that carried it); the half-res map has crossed a second
        # e4m3 tensor (the transposed conv's output feeds the residual units) and the bilinear taps of the transposed conv
        steps = 3 if name == "hm_h" else 2  # (the stem's two convs both write e4m3: two roundings on the way in)
        tol = steps * 0.0625 * np.abs(ref) + 0.01 * np.abs(ref).max()
        bad = np.abs(got - ref) > tol
        assert not bad.any(), (name, int(bad.sum()), float((np.abs(got - ref) / np.maximum(tol, 1e-12)).max()))
    parser = pkg.MPPEHeatmapParser(17, 30, 0.05, 0.5)
    got = parser.to_lists(*parser.decode_batch_device(g_hq, g_hh, [g_tags]))
    counts = []
    worst_cover = 1.0
    FP8_MIN_COVER = 0.8  # measured 0.94: every reference person's detected joints reappear, within 2 px, in ONE fp8 person
    for b in range(B):
        rj, _ = orc.decode(r_hq[b].numpy(), r_hh[b].numpy(), [r_tags[b].numpy()], max_people=30, det_thr=0.05, tag_thr=0.5)
        oj, os_ = orc.decode(g_hq[b].cpu().numpy(), g_hh[b].cpu().numpy(), [g_tags[b].cpu().numpy()], max_people=30, det_thr=0.05, tag_thr=0.5)
        assert np.array_equal(got[b][0], oj) and np.array_equal(got[b][1], os_), b
        counts.append((rj.shape[0], oj.shape[0]))
        # task-level statement that CAN be made on carried values: no person is lost.  Every person the fp32 maps decode to has a
        # counterpart in the fp8 decode -- the fp8 person that shares most of its detected joints within 2 px (the plateaus add
        # duplicate groups around the same blobs, they do not move them)
        for pr in rj:
            det = pr[:, 2] > 0
            if not det.any():
                continue
            best = 0.0
            for po in oj:
                both = det & (po[:, 2] > 0)
                near = both & (np.abs(po[:, 0] - pr[:, 0]) <= 2.0) & (np.abs(po[:, 1] - pr[:, 1]) <= 2.0)
                best = max(best, near.sum() / det.sum())
            worst_cover = min(worst_cover, best)
    print("fp8 chained: groups (fp32 oracle, fp8 engine) per image:", counts, " least share of a reference person's joints found in one fp8 person:", round(worst_cover, 3))
    assert worst_cover >= FP8_MIN_COVER, worst_cover


def test_fp8_requires_calibration_and_taps_track_the_reference(pkg, net_golden):
    net = pkg.HigherHRNet(17, 32, dtype="fp8")
    sd = {k: torch.from_numpy(pkg.synth.synth_param(k, v.shape, 0)) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    x = torch.from_numpy(pkg.synth.synth_images(1, 64, 64, 0)).to(DEV)
    with pytest.raises(pkg._lib.HHError, match="hh_calibrate"):
        net(x)
    net.calibrate(torch.from_numpy(pkg.synth.synth_images(4, 64, 64, 100)).to(DEV))
    net.set_taps(True)
    net(x)
    torch.cuda.synchronize()
    taps = net.read_taps()
    n = 0
    for k in net_golden.files:  # every intermediate tensor of the reference, dequantised with its calibrated scale
        if k.startswith("w32_64/tap/") and k.split("/tap/")[1] in taps and k.split("/tap/")[1] != "deconv#1":
            e = _fp8_close(taps[k.split("/tap/")[1]], net_golden[k], k)
            if "stem#0" in k or "stages.0" in k:  # two to fourteen convs deep (stage 0 keeps its trunk e4m3): a mispacked tap or a
                assert e[1] <= (0.06 if "stem" in k else 0.09), (k, e)  # wrong scale cannot hide here (measured 4.6 % / 7.9 %)
            n += 1
    assert n >= 60
    # new weights invalidate the scales
    net.load_state_dict(sd)
    net.set_taps(False)
    with pytest.raises(pkg._lib.HHError, match="hh_calibrate"):
        net(x)
