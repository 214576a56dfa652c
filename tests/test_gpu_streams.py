"""The handle's stream / thread contract is a property of the LIBRARY (include/sidekit_amd.h, "Conventions"; csrc/xt_api.hip
enter_stream / leave_stream / EntryGuard), not of the caller's discipline.

The reference drives a model from one Python thread on torch's current stream (sidekit/bin/extract_xvectors.py:130-150,
sidekit/nnet/xvector.py:1890-1896) and torch orders everything; this library reuses one set of workspaces from call to call, so two
calls on different streams, or from two threads, would race on them unless the library orders them itself.  Round 4 found one such
race by soak test (a plain forward on the caller's stream against a pipelined batch in slot 0); these tests cover the family."""
import threading

import pytest
import torch

from sidekit_amd.nnet import Xtractor

pytestmark = pytest.mark.gpu


def _wav(g, B, L, dev):
    return 0.1 * torch.randn(B, L, device=dev, generator=g)


def test_calls_from_different_streams_are_ordered_by_the_library(gpu):
    """submit on torch stream A, a small plain forward on stream B (it runs ON stream B in lane 0's workspace, which the pipelined batch of
    slot 0 is still using), collect on A; then the mirror image: an unsplit plain forward on A followed at once by a submit from B (slot 0's
    stream forks from B, which knows nothing of A) -- against a second model that runs one forward at a time on one stream.  Bit for bit."""
    dev = torch.device(gpu)
    a = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=31).to(dev).eval()
    ref = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=31).to(dev).eval()
    a.compute_dtype = ref.compute_dtype = "bf16"
    ref.set_lanes(1)
    g = torch.Generator(device=dev).manual_seed(7)
    big = [_wav(g, 256, 32000, dev) for _ in range(3)]
    small = [_wav(g, B, L, dev) for B, L in ((8, 16000), (1, 64000), (40, 24000))]
    want_big = [ref(w, is_eval=True)[1].clone() for w in big]
    want_small = [ref(w, is_eval=True)[1].clone() for w in small]
    a(big[0], is_eval=True); a.collect(a.submit(big[0])); a(small[0], is_eval=True)      # handles, workspaces, both slots: sized before the streams fork
    torch.cuda.synchronize()
    A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    bad = 0
    for rep in range(12):
        i, j = rep % 3, (rep + 1) % 3
        # (1) pipelined batch from A, plain small forward from B while it runs, collect on A
        with torch.cuda.stream(A):
            t = a.submit(big[i])
        with torch.cuda.stream(B):
            s1 = a(small[j], is_eval=True)[1]
        with torch.cuda.stream(A):
            e1 = a.collect(t)[1]
        # (2) unsplit plain forward on A, submit from B right behind it (same workspace: lane 0 = slot 0 on the first round, slot 1 later)
        with torch.cuda.stream(A):
            s2 = a(small[i], is_eval=True)[1]
        with torch.cuda.stream(B):
            t2 = a.submit(big[j])
            e2 = a.collect(t2)[1]
        # (3) a split (two-lane) plain forward from B while nothing is pending, then features on A
        with torch.cuda.stream(B):
            e3 = a(big[i], is_eval=True)[1]
        torch.cuda.synchronize()
        for got, want in ((e1, want_big[i]), (s1, want_small[j]), (s2, want_small[i]), (e2, want_big[j]), (e3, want_big[i])):
            bad += int(not torch.equal(got, want))
    assert bad == 0, f"{bad} of 60 results differ from the one-at-a-time reference"


def test_submit_on_one_stream_collect_on_another_then_drop_the_outputs(gpu):
    """submit on stream A, collect on stream B, drop the outputs, then allocate-and-fill tensors of the same sizes on A: the caching allocator
    hands A the blocks the dropped outputs occupied (they came from A's pool), so A must have been ordered behind the slot's forward by
    ``collect`` -- else the fill races the forward that is still writing them (``Xtractor.collect``: the submitting stream waits for the slot
    as well, the outputs are ``record_stream``-ed on the collecting one).  A second batch in flight meanwhile must come out as the
    one-at-a-time reference has it, and so must the first one when it is NOT dropped."""
    dev = torch.device(gpu)
    a = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=33).to(dev).eval()
    ref = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=33).to(dev).eval()
    a.compute_dtype = ref.compute_dtype = "bf16"
    ref.set_lanes(1)
    g = torch.Generator(device=dev).manual_seed(9)
    w = [_wav(g, 256, 32000, dev) for _ in range(2)]
    want = [tuple(t.clone() for t in ref(x, is_eval=True)) for x in w]
    a.collect(a.submit(w[0]))
    torch.cuda.synchronize()
    A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    with torch.cuda.stream(A):
        t0 = a.submit(w[0])
        t1 = a.submit(w[1])
    with torch.cuda.stream(B):
        l0, e0 = a.collect(t0)
        keep = (l0.clone(), e0.clone())          # read on B, behind the slot
    shapes = (tuple(l0.shape), tuple(e0.shape))
    del l0, e0, t0                               # the blocks go back to A's pool while batch 1 (and, without the fix, batch 0's tail) may still run
    with torch.cuda.stream(A):
        junk = [torch.full(sh, 7.0, device=dev) for sh in shapes for _ in range(4)]     # same sizes: the allocator reuses the freed blocks
    with torch.cuda.stream(B):
        l1, e1 = a.collect(t1)
    torch.cuda.synchronize()
    assert torch.equal(keep[0], want[0][0]) and torch.equal(keep[1], want[0][1]), "the batch collected on B differs from the reference"
    assert torch.equal(l1, want[1][0]) and torch.equal(e1, want[1][1]), "the second batch differs from the reference"
    assert all(bool((j == 7.0).all()) for j in junk), "a fill on the submitting stream was overwritten by the forward it should have been ordered behind"


def test_two_threads_in_one_handle_are_refused_not_raced(gpu):
    """Two host threads drive ONE model at once (ctypes releases the interpreter lock inside a call, so they really are inside the library
    together): every call either returns the right x-vectors or raises the SK_ESTATE error -- nothing is enqueued by the refused call, and the
    handle keeps working afterwards.  One handle per thread is the supported shape (second half)."""
    dev = torch.device(gpu)
    m = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=32).to(dev).eval()
    m.compute_dtype = "bf16"
    g = torch.Generator(device=dev).manual_seed(8)
    wavs = [_wav(g, 48, 24000, dev), _wav(g, 33, 32000, dev)]
    want = [m(w, is_eval=True)[1].clone() for w in wavs]
    torch.cuda.synchronize()
    results = {0: [], 1: []}
    barrier = threading.Barrier(2)

    def drive(k, model, n):
        st = torch.cuda.Stream(dev)
        barrier.wait()
        for _ in range(n):
            try:
                with torch.cuda.stream(st):
                    e = model(wavs[k], is_eval=True)[1]
                st.synchronize()
                results[k].append(bool(torch.equal(e, want[k])))
            except RuntimeError as exc:
                assert "concurrent entry" in str(exc), exc
                results[k].append("refused")

    th = [threading.Thread(target=drive, args=(k, m, 150)) for k in (0, 1)]
    [t.start() for t in th]
    [t.join() for t in th]
    flat = results[0] + results[1]
    assert len(flat) == 300 and all(r is True or r == "refused" for r in flat), [r for r in flat if r is not True and r != "refused"][:5]
    print(f"concurrent entries refused: {flat.count('refused')} of 300 calls")
    assert torch.equal(m(wavs[0], is_eval=True)[1], want[0])             # the handle is intact
    # one model per thread: no refusals, right answers
    m2 = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=32).to(dev).eval()
    m2.compute_dtype = "bf16"
    m2(wavs[1], is_eval=True)
    torch.cuda.synchronize()
    results = {0: [], 1: []}
    barrier = threading.Barrier(2)
    th = [threading.Thread(target=drive, args=(k, mm, 40)) for k, mm in ((0, m), (1, m2))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert results[0] == [True] * 40 and results[1] == [True] * 40
