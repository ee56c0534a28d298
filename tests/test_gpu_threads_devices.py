"""Per-thread / per-device bookkeeping of the library: workspaces keyed by device and
freed on request, stream ordering between encode_query and the null-stream per-pair calls, the
caller's current device left alone, device buffers checked against the handle's device."""
import threading

import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
torch = pytest.importorskip("torch")
D = qa.DistanceType


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_thread_workspaces_are_stable_and_released():
    rng = np.random.default_rng(0)
    dim = 64
    a = qa.EncodedVectorsU8.encode(rng.random((400_000, dim), dtype=np.float32), qa.VectorParameters(dim, 400_000, D.Dot, False))
    b = qa.EncodedVectorsU8.encode(rng.random((150_000, dim), dtype=np.float32), qa.VectorParameters(dim, 150_000, D.Dot, False))
    q = rng.random(dim, dtype=np.float32)
    qa_, qb = a.encode_query(q), b.encode_query(q)

    def work():
        for enc, qq in ((a, qa_), (b, qb)):  # one thread alternating between two handles
            enc.score_all(qq)                 # host output -> WS_SCORES
            enc.topk(qq, 30)                  # fused path -> WS_FUSED

    work()
    before = free_bytes()
    for _ in range(40):
        work()
    assert abs(free_bytes() - before) < (4 << 20), "alternating handles must not grow device memory"
    qa.thread_release()
    assert free_bytes() >= before + (1 << 20), "qamd_thread_release frees the cached score workspace"
    work()  # and everything still works afterwards

    # a short-lived thread gives back what it cached when it ends
    base = free_bytes()
    t = threading.Thread(target=work)
    t.start()
    t.join()
    assert abs(free_bytes() - base) < (4 << 20)


def test_per_pair_calls_are_ordered_after_encode_query_on_a_side_stream(qo):
    rng = np.random.default_rng(1)
    n, dim = 2000, 768
    data = rng.random((n, dim), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(data, qa.VectorParameters(dim, n, D.Dot, False))
    rows, meta = qo.u8_encode(data, qo.DOT, False)
    side = torch.cuda.Stream()  # non-blocking w.r.t. the null stream
    filler = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    qobj = None
    for trial in range(20):
        query = rng.random(dim, dtype=np.float32)
        dq = torch.from_numpy(query).cuda()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(4):
                filler.add_(1.0)  # keep the side stream busy so the encode kernel starts late
            qobj = enc.encode_query(dq, reuse=qobj)
        got = enc.score_point(qobj, trial)  # null stream: must wait for the encode above
        codes, qoff = qo.u8_encode_query(meta, query)
        want = qo.u8_score_point(meta, rows, codes, qoff, trial, order=qo.ORDER_AVX2)
        assert np.float32(got).view(np.uint32) == np.float32(want).view(np.uint32), trial
        assert np.array_equal(qobj.encoded_query, codes)
    torch.cuda.synchronize()


def test_calls_leave_the_current_device_and_check_buffer_devices():
    rng = np.random.default_rng(2)
    data = rng.random((1000, 32), dtype=np.float32)
    enc = qa.EncodedVectorsU8.encode(torch.from_numpy(data).cuda(), qa.VectorParameters(32, 1000, D.Dot, False))
    assert enc.device == 0 and torch.cuda.current_device() == 0
    q = enc.encode_query(data[0])
    out = torch.empty(1000, dtype=torch.float32, device="cuda:0")
    enc.score_all(q, out=out)
    assert_bits_equal(out.cpu().numpy(), enc.score_all(q), "device vs host output")
    if torch.cuda.device_count() > 1:
        other = torch.empty(1000, dtype=torch.float32, device="cuda:1")
        with pytest.raises(ValueError, match="cuda:1"):
            enc.score_all(q, out=other)
        with torch.cuda.device(1):
            enc.score_all(q, out=out)  # runs on the handle's device 0 ...
            assert torch.cuda.current_device() == 1  # ... and leaves the caller's device alone
        far = qa.EncodedVectorsU8.encode(torch.from_numpy(data).to("cuda:1"), qa.VectorParameters(32, 1000, D.Dot, False))
        assert far.device == 1
        assert_bits_equal(far.score_all(far.encode_query(data[0])), enc.score_all(q), "store on cuda:1")
        for _ in range(10):  # one thread alternating between two DEVICES: per-device workspaces
            enc.score_all(q)
            far.score_all(far.encode_query(data[0]))


def test_batched_topk_from_several_threads_on_one_handle():
    """Search threads share one store (the reference's scorer is `&self`): concurrent topk_batch calls of
    different sizes - row-streaming, several tiles, query-streaming - on their own streams, the first of
    them racing to gather the handle's pivot sample; every list equals the serial result."""
    rng = np.random.default_rng(5)
    n, dim = 120_000, 256
    enc = qa.EncodedVectorsU8.encode(rng.random((n, dim), dtype=np.float32), qa.VectorParameters(dim, n, D.Dot, False))
    sizes = [7, 100, 300, 1000]
    queries = [rng.random((q, dim), dtype=np.float32) for q in sizes]
    results = [None] * len(sizes)
    errors = []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):
                    results[i] = enc.topk_batch(enc.encode_query_batch(queries[i]), 20)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
        finally:
            qa.thread_release()

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(sizes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, qs in enumerate(queries):
        ids, sc = enc.topk_batch(enc.encode_query_batch(qs), 20)
        assert np.array_equal(results[i][0], ids), sizes[i]
        assert_bits_equal(results[i][1], sc, f"{sizes[i]} queries, concurrent vs serial")
        wi, ws = enc.topk(enc.encode_query(qs[0]), 20)
        assert np.array_equal(ids[0], wi) and np.array_equal(sc[0].view(np.uint32), ws.view(np.uint32))
