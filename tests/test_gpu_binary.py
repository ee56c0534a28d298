"""HIP binary path vs the oracle (bit-exact) and the reference's own known-answer spec
(quantization/tests/test_binary.rs)."""
import numpy as np
import pytest

from util import assert_bits_equal

pytestmark = pytest.mark.gpu

qa = pytest.importorskip("quantization_amd")
D = qa.DistanceType
S = qa.BitsStoreType

DIMS_U8 = [0, 1, 8, 33, 65, 3 * 129, 1024, 128, 64, 32, 129, 2048 + 17]
DIMS_U128 = [1, 3 * 129, 1024]


def _pm1(n, dim, seed=42):
    """test_binary.rs:14-25: vectors of exact +-1.0."""
    rng = np.random.default_rng(seed)
    v = np.where(rng.random((n, dim)) < 0.5, -1.0, 1.0).astype(np.float32)
    q = np.where(rng.random(dim) < 0.5, -1.0, 1.0).astype(np.float32)
    return v, q


def _cases():
    return [(d, S.U8) for d in DIMS_U8] + [(d, S.U128) for d in DIMS_U128]


@pytest.mark.parametrize("dim,store", _cases())
@pytest.mark.parametrize("dist", [D.Dot, D.L1, D.L2])
@pytest.mark.parametrize("invert", [False, True])
def test_binary_bit_exact(qo, dim, store, dist, invert):
    n = 128
    data, query = _pm1(n, dim)
    vp = qa.VectorParameters(dim, n, dist, invert)
    enc = qa.EncodedVectorsBin.encode(data, vp, store=store)
    rows = qo.bin_encode(data, int(store))
    assert qa.EncodedVectorsBin.get_quantized_vector_size_from_params(vp, store) == qo.bin_row_bytes(dim, int(store))
    assert np.array_equal(enc.storage_bytes(), rows), "packed rows differ"
    q = enc.encode_query(query)
    qbits = qo.bin_encode(query[None, :], int(store))[0]
    assert np.array_equal(q.encoded_vector, qbits)
    want = qo.bin_score_all(rows, qbits, dim, int(dist), invert, int(store))
    assert_bits_equal(enc.score_all(q), want, "score_all")
    if qo.ref() is not None:
        assert_bits_equal(enc.score_all(q), qo.bin_score_all(rows, qbits, dim, int(dist), invert, int(store),
                                                              use_ref=True), "vs _ref popcount")
    for i in (0, 77, n - 1):
        assert_bits_equal([enc.score_point(q, i)], [want[i]], "score_point")
        assert_bits_equal([enc.score_internal(i, (i * 7) % n)],
                          [qo.bin_score_internal(rows, dim, int(dist), invert, i, (i * 7) % n, int(store))],
                          "score_internal")
    ids = np.array([5, 5, 127, 0], dtype=np.uint32)
    assert_bits_equal(enc.score_ids(q, ids), want[ids], "score_ids")


@pytest.mark.parametrize("dim,store", _cases())
@pytest.mark.parametrize("invert", [False, True])
def test_binary_dot_known_answer(qo, dim, store, invert):
    """test_binary.rs:39-71: on +-1 vectors the Dot score IS the f32 dot product."""
    data, query = _pm1(128, dim)
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, 128, D.Dot, invert), store=store)
    got = enc.score_all(enc.encode_query(query))
    want = (data @ query).astype(np.float32)
    assert np.array_equal(got, -want if invert else want)


@pytest.mark.parametrize("dist", [D.L1, D.L2])
@pytest.mark.parametrize("invert", [False, True])
def test_binary_l1_l2_ordering(qo, dist, invert):
    """test_binary.rs:243-263: sorted order equals the true metric's order."""
    for dim in (8, 33, 65, 387):
        data, query = _pm1(128, dim)
        enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(dim, 128, dist, invert))
        got = enc.score_all(enc.encode_query(query))
        true = np.array([qo.metric_f32(int(dist), query, data[i]) for i in range(128)])
        if invert:
            true = -true
        # ties are possible: compare the score VALUES along both orders
        assert np.array_equal(got[np.argsort(true, kind="stable")], np.sort(got)), dim


def test_binary_real_valued_inputs_and_edge_values(qo):
    rng = np.random.default_rng(3)
    data = rng.standard_normal((300, 200)).astype(np.float32)
    data[0, :6] = [0.0, -0.0, np.nan, np.inf, -np.inf, 1e-45]
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(200, 300, D.Dot, False))
    assert np.array_equal(enc.storage_bytes(), qo.bin_encode(data))


def test_binary_save_load_and_device_inputs(qo, tmp_path):
    import json
    torch = pytest.importorskip("torch")
    n, dim = 1000, 1024
    data, query = _pm1(n, dim, seed=5)
    vp = qa.VectorParameters(dim, n, D.Dot, False)
    enc = qa.EncodedVectorsBin.encode(torch.from_numpy(data).cuda(), vp)
    rows = qo.bin_encode(data)
    assert np.array_equal(enc.storage_bytes(), rows)
    q = enc.encode_query(torch.from_numpy(query).cuda())
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    enc.score_all(q, out=out)
    torch.cuda.synchronize()
    want = qo.bin_score_all(rows, qo.bin_encode(query[None])[0], dim, qo.DOT, False)
    assert_bits_equal(out.cpu().numpy(), want, "device scores")
    enc.save(tmp_path / "b.bin", tmp_path / "b.json")
    assert open(tmp_path / "b.bin", "rb").read() == rows.tobytes()
    assert json.load(open(tmp_path / "b.json")) == {
        "vector_parameters": {"dim": dim, "count": n, "distance_type": "Dot", "invert": False}}
    back = qa.EncodedVectorsBin.load(tmp_path / "b.bin", tmp_path / "b.json", vp)
    assert_bits_equal(back.score_all(back.encode_query(query)), want, "reloaded")
    with pytest.raises(OSError):
        qa.EncodedVectorsBin.load(tmp_path / "b.bin", tmp_path / "b.json", qa.VectorParameters(dim, n - 1, D.Dot, False))


def test_binary_stop_and_errors():
    data = np.ones((1000, 64), dtype=np.float32)
    with pytest.raises(qa.EncodingError) as e:
        qa.EncodedVectorsBin.encode(data, qa.VectorParameters(64, 1000, D.Dot, False), stop_condition=lambda: True)
    assert e.value.stopped
    enc = qa.EncodedVectorsBin.encode(data, qa.VectorParameters(64, 1000, D.Dot, False))
    q = enc.encode_query(data[0])
    with pytest.raises(IndexError):
        enc.score_point(q, 1000)
