"""What does a cross-stream edge cost on this box?  (GPU box)  python tools/probes/event_cost.py
A chain of N ~15 us kernels on stream A, timed end to end:
  plain          -- nothing between the kernels;
  wait-done      -- before every kernel, hipStreamWaitEvent on an event of stream B that completed long ago;
  record         -- after every kernel, hipEventRecord (nobody waits);
  ping-pong      -- kernels alternate between A and B, each waiting for the previous one's event (one real hop per kernel);
  wait-late1/2   -- every fifth kernel of A is followed by a wait for one / two events of other streams that were pending when the wait was
                    enqueued and are complete long before A gets there (per-kernel figure: a fifth of the wait's cost);
  wait-late1-rec -- the same with one wait + one record on A;
  ping-pong-mem  -- the same hops through hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory instead of events.
"""
import ctypes
import time
import torch

dev = torch.device("cuda:0")
x = torch.randn(8 << 20, device=dev)
y = torch.empty_like(x)
A, B, Cs = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def k(s):
    with torch.cuda.stream(s):
        torch.mul(x, 1.0001, out=y)


hip = ctypes.CDLL("libamdhip64.so")
sig = ctypes.c_void_p()
assert hip.hipExtMallocWithFlags(ctypes.byref(sig), 8, 0x2) == 0  # hipMallocSignalMemory
hip.hipStreamWriteValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint]
hip.hipStreamWaitValue32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint, ctypes.c_uint32]
counter = [0]


def run(mode):
    done = torch.cuda.Event()
    with torch.cuda.stream(B):
        k(B)
        done.record(B)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event() for _ in range(N + 1)]
    evs2 = [torch.cuda.Event() for _ in range(N + 1)]
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(A)
    for i in range(N):
        if mode == "plain":
            k(A)
        elif mode == "wait-done":
            A.wait_event(done)
            k(A)
        elif mode == "record":
            k(A)
            evs[i].record(A)
        elif mode == "ping-pong":
            s = A if i % 2 == 0 else B
            if i:
                s.wait_event(evs[i - 1])
            k(s)
            evs[i].record(s)
        elif mode in ("wait-late1", "wait-late2", "wait-late1-rec"):
            # the events are recorded on B / C behind a short kernel enqueued just now: pending at enqueue time (so the waits become real
            # barrier packets), complete long before stream A reaches them (A has four kernels queued in front)
            if i % 5 == 0:
                k(B)
                evs[i].record(B)
                if mode == "wait-late2":
                    k(Cs)
                    evs2[i].record(Cs)
            k(A)
            if i % 5 == 4:
                A.wait_event(evs[i - 4])
                if mode == "wait-late2":
                    A.wait_event(evs2[i - 4])
                if mode == "wait-late1-rec":
                    evs2[i].record(A)
        elif mode == "ping-pong-mem":
            s = A if i % 2 == 0 else B
            if i:
                assert hip.hipStreamWaitValue32(s.cuda_stream, sig, counter[0], 0, 0xffffffff) == 0  # >= counter
            k(s)
            counter[0] += 1
            assert hip.hipStreamWriteValue32(s.cuda_stream, sig, counter[0], 0) == 0
    if mode == "ping-pong":
        A.wait_event(evs[N - 1])
    if mode == "ping-pong-mem":
        assert hip.hipStreamWaitValue32(A.cuda_stream, sig, counter[0], 0, 0xffffffff) == 0
    e1.record(A)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


for mode in ("plain", "wait-done", "record", "ping-pong", "ping-pong-mem", "wait-late1", "wait-late2", "wait-late1-rec", "plain"):
    run(mode)
    print(f"{mode:10s} {min(run(mode) for _ in range(3)):7.2f} us per kernel", flush=True)
