#!/usr/bin/env python3
"""Build-time check behind the hand-counted LDS waits of basicblock_fused_pc.hip.

The kernel issues its fragment reads as asm `ds_read_b128` and waits with `s_waitcnt lgkmcnt(N)`, N > 0, counting on LDS
operations completing in order.  Scalar memory loads (s_load / s_buffer_load) share that counter and return OUT of order: one
of them in flight at such a wait could let a fragment be used before it has arrived.  The source keeps every kernel-argument
read outside the tile loop; this script checks that the COMPILER kept it so in the gfx950 code it emitted: it generates the
device assembly (hipcc -S, seconds) and walks it twice in program order (the second walk stands for loop back edges): between an
s_load / s_buffer_load and the next `lgkmcnt(0)` no `lgkmcnt(N > 0)` may occur.

    python3 tools/check_lds_wait_isa.py [source.hip ...]      exit code 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "pytorch-human-pose_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def device_asm(src: str) -> str:
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-I", CSRC, src, "-o", out],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


def check(asm: str):
    """-> list of (function, line number, text) of partial LDS waits with a scalar load possibly in flight"""
    bad = []
    func = None
    body = []
    for ln in asm.splitlines():
        m = re.match(r"^(\w+):\s*; @", ln)
        if m:
            func, body = m.group(1), []
            continue
        if func is None:
            continue
        body.append(ln)
        if ln.strip().startswith("s_endpgm"):
            smem = False
            for rnd in range(2):  # second walk: the state a loop back edge carries into its head
                for i, t in enumerate(body):
                    t = t.strip()
                    if t.startswith(("s_load_", "s_buffer_load_")):
                        smem = True
                    w = re.search(r"lgkmcnt\((\d+)\)", t)
                    if t.startswith("s_waitcnt") and w:
                        if int(w.group(1)) == 0:
                            smem = False
                        elif smem and rnd == 1 or (smem and rnd == 0):
                            bad.append((func, i, t))
            func = None
    return sorted(set(bad))


if __name__ == "__main__":
    srcs = sys.argv[1:] or [os.path.join(CSRC, "basicblock_fused_pc.hip")]
    rc = 0
    for s in srcs:
        asm = device_asm(s)
        nwait = len(re.findall(r"s_waitcnt[^\n]*lgkmcnt\([1-9]", asm))
        bad = check(asm)
        print(f"{os.path.basename(s)}: {nwait} partial lgkmcnt waits, {len(bad)} with a scalar load possibly in flight")
        for f, i, t in bad[:10]:
            print("   ", f, i, t)
        rc |= bool(bad)
    sys.exit(rc)
