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
