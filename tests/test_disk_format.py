"""The on-disk boundary: `load` must accept metadata files this library did NOT write - serde_json's compact
output with ryu's float formatting (tests/golden/meta_*.json, written by tests/golden/make_meta_fixtures.py's independent
formatter; quantization/src/encoded_vectors_u8.rs:24-31,263-288, encoded_vectors_pq.rs:39-44,498-523,
encoded_vectors_binary.rs:21-24,260-286, encoded_storage.rs:32-59) - and must reject what serde_json::from_str rejects.
Malformed metadata fails before anything touches a GPU, so those tests run everywhere; loading the fixtures and scoring
them bit-exactly against the oracle needs the GPU."""
import json
import os
import shutil

import numpy as np
import pytest

from util import assert_bits_equal

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
qa = pytest.importorskip("quantization_amd")
D = qa.DistanceType


def _u8_vp():
    return qa.VectorParameters(20, 7, D.L2, False)


# ------------------------------------------------------------------------------------------------ rejected metadata (CPU)
GOOD_U8 = open(os.path.join(GOLD, "meta_u8.json")).read()
BAD_U8 = {
    "truncated file": GOOD_U8[:-9],
    "trailing characters": GOOD_U8 + " x",
    "missing field": GOOD_U8.replace('"alpha":0.01751986,', ""),
    "duplicate field": GOOD_U8.replace('"offset":', '"alpha":1.0,"offset":'),
    "null where an f32 belongs (serde_json writes NaN as null and cannot read it back)": GOOD_U8.replace("0.01751986", "null"),
    "string where a number belongs": GOOD_U8.replace("0.01751986", '"0.01751986"'),
    "fraction in a usize": GOOD_U8.replace('"actual_dim":32', '"actual_dim":32.0'),
    "negative usize": GOOD_U8.replace('"count":7', '"count":-7'),
    "unknown distance_type": GOOD_U8.replace('"L2"', '"Cosine"'),
    "distance_type not a string": GOOD_U8.replace('"L2"', "2"),
    "invert not a boolean": GOOD_U8.replace("false", "0"),
    "leading zero": GOOD_U8.replace('"dim":20', '"dim":020'),
    "bare fraction": GOOD_U8.replace("0.01751986", ".01751986"),
    "NaN literal": GOOD_U8.replace("0.01751986", "NaN"),
    "f64 overflow": GOOD_U8.replace("0.01751986", "1e400"),
    "single quotes": GOOD_U8.replace('"', "'"),
    "vector_parameters not a struct": GOOD_U8.replace('{"dim":20,"count":7,"distance_type":"L2","invert":false}', "[20,7]"),
    "an array at the top": "[" + GOOD_U8 + "]",
    "empty file": "",
}


@pytest.mark.parametrize("what", sorted(BAD_U8))
def test_u8_load_rejects_what_serde_json_rejects(tmp_path, what):
    meta = tmp_path / "meta.json"
    meta.write_text(BAD_U8[what])
    with pytest.raises(OSError):  # std::io::Error in the reference (serde_json::Error converts into it)
        qa.EncodedVectorsU8.load(os.path.join(GOLD, "rows_u8.bin"), str(meta), _u8_vp())


def test_load_reports_a_missing_file_as_io_error(tmp_path):
    with pytest.raises(OSError):
        qa.EncodedVectorsU8.load(os.path.join(GOLD, "rows_u8.bin"), str(tmp_path / "nope.json"), _u8_vp())
    with pytest.raises(OSError):
        qa.EncodedVectorsBin.load(str(tmp_path / "nope.bin"), os.path.join(GOLD, "meta_bin.json"),
                                  qa.VectorParameters(70, 5, D.L1, True))


@pytest.mark.parametrize("edit", ["centroid row too short", "255 centroids", "division does not tile", "division start is a float",
                                  "null centroid value"])
def test_pq_load_rejects_inconsistent_metadata(tmp_path, edit):
    js = json.loads(open(os.path.join(GOLD, "meta_pq.json")).read())
    if edit == "centroid row too short":
        js["centroids"][17] = js["centroids"][17][:5]
    elif edit == "255 centroids":
        js["centroids"] = js["centroids"][:255]
    elif edit == "division does not tile":
        js["vector_division"][1] = {"start": 3, "end": 6}
    elif edit == "division start is a float":
        js["vector_division"][0] = {"start": 0.0, "end": 4}
    else:
        js["centroids"][3][2] = None
    meta = tmp_path / "meta.json"
    meta.write_text(json.dumps(js))
    with pytest.raises(OSError):
        qa.EncodedVectorsPQ.load(os.path.join(GOLD, "rows_pq.bin"), str(meta), qa.VectorParameters(6, 9, D.Dot, False))


# ------------------------------------------------------------------------------------------------ accepted files (GPU)
def _parsed_u8_meta(path):
    js = json.loads(open(path).read())
    f = lambda k: np.float32(js[k])  # json -> f64 -> f32: serde_json's own path for an f32 field
    return js, f("alpha"), f("offset"), f("multiplier")


@pytest.mark.gpu
@pytest.mark.parametrize("meta_file", ["meta_u8.json", "meta_u8_pretty.json"])
def test_u8_loads_a_reference_format_store_it_did_not_write(qo, meta_file):
    enc = qa.EncodedVectorsU8.load(os.path.join(GOLD, "rows_u8.bin"), os.path.join(GOLD, meta_file), _u8_vp())
    js, alpha, offset, mult = _parsed_u8_meta(os.path.join(GOLD, meta_file))
    md = enc.metadata
    for got, want in ((md["alpha"], alpha), (md["offset"], offset), (md["multiplier"], mult)):
        assert np.float32(got).view(np.uint32) == want.view(np.uint32)
    assert md["actual_dim"] == 32 and md["vector_parameters"].distance_type == D.L2 and not md["vector_parameters"].invert
    data = np.load(os.path.join(GOLD, "data_u8.npy"))
    rows, meta = qo.u8_encode(data, qo.L2, False)  # the oracle's own metadata: the file's text reads back to the same bits
    assert (np.float32(meta.alpha).view(np.uint32), np.float32(meta.offset).view(np.uint32),
            np.float32(meta.multiplier).view(np.uint32)) == (alpha.view(np.uint32), offset.view(np.uint32), mult.view(np.uint32))
    assert np.array_equal(enc.storage_bytes(), rows)
    assert np.array_equal(np.fromfile(os.path.join(GOLD, "rows_u8.bin"), dtype=np.uint8).reshape(rows.shape), rows)
    q = np.linspace(-0.5, 1.2, 20, dtype=np.float32)
    codes, qoff = qo.u8_encode_query(meta, q)
    assert_bits_equal(enc.score_all(enc.encode_query(q)), qo.u8_score_all(meta, rows, codes, qoff, order=qo.ORDER_AVX2),
                      "scores of the loaded store")


@pytest.mark.gpu
def test_u8_loads_the_degenerate_interval_and_negative_zero(qo):
    enc = qa.EncodedVectorsU8.load(os.path.join(GOLD, "rows_u8_zero.bin"), os.path.join(GOLD, "meta_u8_zero.json"),
                                   qa.VectorParameters(16, 3, D.Dot, True))
    md = enc.metadata
    assert np.float32(md["alpha"]).view(np.uint32) == 0 and np.float32(md["offset"]).view(np.uint32) == 0x80000000
    assert np.float32(md["multiplier"]).view(np.uint32) == 0x80000000  # "-0.0" keeps its sign


@pytest.mark.gpu
def test_pq_loads_a_reference_format_store_it_did_not_write(qo):
    vp = qa.VectorParameters(6, 9, D.Dot, False)
    enc = qa.EncodedVectorsPQ.load(os.path.join(GOLD, "rows_pq.bin"), os.path.join(GOLD, "meta_pq.json"), vp)
    cen = np.load(os.path.join(GOLD, "centroids_pq.npy"))
    assert_bits_equal(enc.centroids, cen, "centroids parsed from ryu-formatted text (1e-7, -0.0, 1e-45, 3.4028235e38, ...)")
    assert [(r.start, r.stop) for r in enc.vector_division] == [(0, 4), (4, 6)]
    rows = np.fromfile(os.path.join(GOLD, "rows_pq.bin"), dtype=np.uint8).reshape(9, 2)
    assert np.array_equal(enc.storage_bytes(), rows)
    q = np.linspace(-1, 1, 6, dtype=np.float32)
    lut = qo.pq_encode_query(q, 4, cen, qo.DOT, False)
    assert_bits_equal(enc.score_all(enc.encode_query(q)), qo.pq_score_all(rows, lut, order=qo.ORDER_SSE), "PQ scores")


@pytest.mark.gpu
def test_binary_loads_a_reference_format_store_it_did_not_write(qo):
    vp = qa.VectorParameters(70, 5, D.L1, True)
    enc = qa.EncodedVectorsBin.load(os.path.join(GOLD, "rows_bin.bin"), os.path.join(GOLD, "meta_bin.json"), vp)
    rows = np.fromfile(os.path.join(GOLD, "rows_bin.bin"), dtype=np.uint8).reshape(5, -1)
    assert np.array_equal(enc.storage_bytes(), rows)
    q = np.where(np.arange(70) % 3 == 0, -1.0, 1.0).astype(np.float32)
    want = qo.bin_score_all(rows, qo.bin_encode(q[None, :])[0], 70, qo.L1, True)
    assert_bits_equal(enc.score_all(enc.encode_query(q)), want, "binary scores")


@pytest.mark.gpu
def test_load_checks_the_row_file_against_the_callers_count(tmp_path):
    """encoded_storage.rs:40-51: file length must be quantized_vector_size * count of the CALLER's parameters."""
    for bad_vp in (qa.VectorParameters(20, 8, D.L2, False), qa.VectorParameters(36, 7, D.L2, False)):
        with pytest.raises(OSError, match="Loaded storage size 252 is not equal to expected size"):
            qa.EncodedVectorsU8.load(os.path.join(GOLD, "rows_u8.bin"), os.path.join(GOLD, "meta_u8.json"), bad_vp)
    short = tmp_path / "rows.bin"
    shutil.copy(os.path.join(GOLD, "rows_u8.bin"), short)
    with open(short, "r+b") as f:
        f.truncate(251)
    with pytest.raises(OSError, match="Loaded storage size 251"):
        qa.EncodedVectorsU8.load(str(short), os.path.join(GOLD, "meta_u8.json"), _u8_vp())


@pytest.mark.gpu
def test_what_the_library_writes_is_the_fixture_text(tmp_path, qo):
    """save() of the loaded fixture store reproduces the fixture's metadata bytes: the writer's float formatting is ryu's."""
    enc = qa.EncodedVectorsU8.load(os.path.join(GOLD, "rows_u8.bin"), os.path.join(GOLD, "meta_u8.json"), _u8_vp())
    enc.save(str(tmp_path / "d.bin"), str(tmp_path / "m.json"))
    assert open(tmp_path / "m.json").read() == GOOD_U8
    assert open(tmp_path / "d.bin", "rb").read() == open(os.path.join(GOLD, "rows_u8.bin"), "rb").read()
    pq = qa.EncodedVectorsPQ.load(os.path.join(GOLD, "rows_pq.bin"), os.path.join(GOLD, "meta_pq.json"),
                                  qa.VectorParameters(6, 9, D.Dot, False))
    pq.save(str(tmp_path / "pd.bin"), str(tmp_path / "pm.json"))
    assert open(tmp_path / "pm.json").read() == open(os.path.join(GOLD, "meta_pq.json")).read()
