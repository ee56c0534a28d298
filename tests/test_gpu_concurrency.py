"""Threading and stream semantics promised by include/quantization_amd.h: score_* calls are
re-entrant on a shared handle (the reference's `&self` scorers are called from many search
threads), and with device buffers they only enqueue work — so they can be captured
into a HIP graph."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
torch = pytest.importorskip("torch")
D = qa.DistanceType


def test_concurrent_scoring_on_one_handle():
    rng = np.random.default_rng(0)
    n, dim, nthreads = 200_000, 256, 6
    data = rng.random((n, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    queries = rng.random((nthreads, dim), dtype=np.float32)
    want = [enc.score_all(enc.encode_query(q)) for q in queries]
    want_top = [enc.topk(enc.encode_query(q), 20) for q in queries]
    errors = []

    def worker(i):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                for _ in range(10):
                    q = enc.encode_query(queries[i])
                    got = enc.score_all(q)
                    if not np.array_equal(got.view(np.uint32), want[i].view(np.uint32)):
                        errors.append(f"thread {i}: score_all differs")
                    ids, sc = enc.topk(q, 20)
                    if not (np.array_equal(ids, want_top[i][0]) and np.array_equal(sc, want_top[i][1])):
                        errors.append(f"thread {i}: topk differs")
                    if enc.score_point(q, i) != want[i][i]:
                        errors.append(f"thread {i}: score_point differs")
        except Exception as e:  # pragma: no cover
            errors.append(f"thread {i}: {e!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(nthreads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_scan_is_capturable_in_a_hip_graph():
    """encode_query (device query, re-used object) + score_all (device output) allocate nothing
    and never synchronise: a query loop can be replayed as one hipGraph."""
    rng = np.random.default_rng(1)
    n, dim = 100_000, 768
    data = torch.from_numpy(rng.random((n, dim), dtype=np.float32)).cuda()
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.L2, False))
    q_dev = torch.zeros(dim, device="cuda")
    out = torch.empty(n, device="cuda")
    qobj = enc.encode_query(q_dev)  # allocate the query object outside the capture
    benc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    bq = benc.encode_query(q_dev)
    bout = torch.empty(n, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        enc.encode_query(q_dev, reuse=qobj)
        enc.score_all(qobj, out=out)
        benc.encode_query(q_dev, reuse=bq)
        benc.score_all(bq, out=bout)
    for seed in (5, 6):
        q_host = np.random.default_rng(seed).random(dim, dtype=np.float32) - 0.5
        q_dev.copy_(torch.from_numpy(q_host))
        g.replay()
        torch.cuda.synchronize()
        want = enc.score_all(enc.encode_query(q_host))
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
        assert np.array_equal(bout.cpu().numpy(), benc.score_all(benc.encode_query(q_host)))


def test_first_calls_from_two_threads_at_once_see_the_kernel_set_up():
    """Per-device one-time set-up (hipFuncSetAttribute: the opt-in to more than 64 KiB of dynamic LDS) is two-phase
    (csrc/common.hpp DeviceOnce): a thread that arrives while another is still inside the set-up waits for it instead of
    launching with the attribute not yet applied.  A FRESH process, so that the calls below really are the first ones: two
    threads start together on the first PQ m = 96 scan (144 KiB of LDS) and the first 129-query u8 batch (the
    queries-in-registers MFMA kernel, 160 KiB) of the process; every call must succeed and give the single-threaded bits."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, threading
sys.path.insert(0, %r)
import numpy as np
import quantization_amd as qa
D = qa.DistanceType
rng = np.random.default_rng(1)
n = 20000
rows = rng.integers(0, 256, size=(n, 96), dtype=np.uint8)
cen = rng.random((256, 768), dtype=np.float32)
pq = qa.EncodedVectorsPQ.from_storage(rows, qa.VectorParameters(768, n, D.Dot, False), 8, cen)
u8 = qa.EncodedVectorsU8.encode(rng.random((140000, 768), dtype=np.float32), qa.VectorParameters(768, 140000, D.Dot, False))
pq_q = [pq.encode_query(rng.random(768, dtype=np.float32)) for _ in range(4)]
batches = [u8.encode_query_batch(rng.random((129, 768), dtype=np.float32)) for _ in range(4)]
start = threading.Barrier(4)
out, errs = {}, []
def run(i):
    try:
        start.wait()
        out[i] = (pq.score_all(pq_q[i]), u8.topk_batch(batches[i], 10)) if i %% 2 == 0 else \
                 (u8.topk_batch(batches[i], 10), pq.score_all(pq_q[i]))
    except Exception as e:
        errs.append(repr(e))
ts = [threading.Thread(target=run, args=(i,)) for i in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
assert not errs, errs
for i in range(4):
    a, b = out[i] if i %% 2 == 0 else out[i][::-1]
    assert np.array_equal(a.view(np.uint32), pq.score_all(pq_q[i]).view(np.uint32))
    ids, sc = u8.topk_batch(batches[i], 10)
    assert np.array_equal(b[0], ids) and np.array_equal(b[1].view(np.uint32), sc.view(np.uint32))
print("OK")
""" % root
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "OK" in res.stdout, (res.stdout + res.stderr)[-3000:]
