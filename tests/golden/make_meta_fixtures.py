#!/usr/bin/env python3
"""Writes the on-disk-format fixtures under tests/golden/: metadata files in serde_json's compact form with ryu's float
formatting - what `serde_json::to_vec(&Metadata)` emits (quantization/src/encoded_vectors_u8.rs:24-31,263-271,
encoded_vectors_pq.rs:39-44,498-506, encoded_vectors_binary.rs:21-24,260-268) - plus the raw row files
(`EncodedStorage::save_to_file`, encoded_storage.rs:53-58) the oracle encodes for the same data.  The formatter below is an
independent restatement of ryu's `pretty` layout for f32 (digits: the shortest decimal that reads back as the same f32;
layout: 12340000000.0 / 12.34 / 0.001234 for -6 < exponent <= 13, else 1e30 / 1.234e33), NOT the library's writer: the
library must load these files although it did not write them.

    python tests/golden/make_meta_fixtures.py        (needs the oracle; run in the build container)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import qoracle as qo  # noqa: E402


def ryu_f32(v) -> str:
    v = np.float32(v)
    if not np.isfinite(v):
        return "null"  # serde_json writes null for NaN / inf
    sign = "-" if np.signbit(v) else ""
    if v == 0:
        return sign + "0.0"
    digits, exp10 = f"{abs(float(v)):.8e}".split("e")  # start from 9 significant digits, then shorten
    for p in range(0, 9):
        t = f"{abs(float(v)):.{p}e}"
        if np.float32(float(t)) == abs(v):
            digits, exp10 = t.split("e")
            break
    d = digits.replace(".", "")
    e = int(exp10)          # value = d[0].d[1:] * 10^e
    length = len(d)
    kk = e + 1              # 10^(kk-1) <= v < 10^kk
    k = kk - length         # value = d * 10^k
    if 0 <= k and kk <= 13:
        return sign + d + "0" * k + ".0"
    if 0 < kk <= 13:
        return sign + d[:kk] + "." + d[kk:]
    if -6 < kk <= 0:
        return sign + "0." + "0" * (-kk) + d
    if length == 1:
        return sign + d + "e" + str(kk - 1)
    return sign + d[0] + "." + d[1:] + "e" + str(kk - 1)


def vp_json(dim, count, dist, invert):
    return '{"dim":%d,"count":%d,"distance_type":"%s","invert":%s}' % (dim, count, dist, "true" if invert else "false")


def main():
    assert ryu_f32(1e-7) == "1e-7" and ryu_f32(1.0) == "1.0" and ryu_f32(-0.0) == "-0.0" and ryu_f32(0.3) == "0.3"
    assert ryu_f32(1.17549435e-38) == "1.1754944e-38" and ryu_f32(16777216.0) == "16777216.0" and ryu_f32(1e13) == "1e13"
    assert ryu_f32(0.000001) == "0.000001" and ryu_f32(123456.79) == "123456.79" and ryu_f32(1.5e-6) == "0.0000015"
    rng = np.random.default_rng(2026)

    # scalar u8: 7 x 20 (actual_dim 32), L2; the data spans [-0.75, 1.5] so that offset is negative
    data = (rng.random((7, 20), dtype=np.float32) * np.float32(2.25) - np.float32(0.75)).astype(np.float32)
    rows, meta = qo.u8_encode(data, qo.L2, False)
    open(os.path.join(HERE, "rows_u8.bin"), "wb").write(rows.tobytes())
    np.save(os.path.join(HERE, "data_u8.npy"), data)
    js = '{"actual_dim":%d,"alpha":%s,"offset":%s,"multiplier":%s,"vector_parameters":%s}' % (
        meta.actual_dim, ryu_f32(meta.alpha), ryu_f32(meta.offset), ryu_f32(meta.multiplier), vp_json(20, 7, "L2", False))
    open(os.path.join(HERE, "meta_u8.json"), "w").write(js)
    # the same metadata as serde_json::to_string_pretty would lay it out, with the keys in another order, an unknown field
    # (serde skips it) and the floats in other valid JSON spellings of the same f32 values
    alt = lambda v: f"{float(np.float32(v)):.9E}".replace("E-0", "E-").replace("E+0", "E+")
    pretty = ('{\n  "vector_parameters": {\n    "invert": false,\n    "distance_type": "L2",\n    "count": 7,\n    "dim": 20\n  },\n'
              '  "format_note": {"written_by": ["a test", 1, null, true], "escaped \\"key\\"": "\\u00e9\\n"},\n'
              '  "multiplier": %s,\n  "offset": %s,\n  "alpha": %s,\n  "actual_dim": 32\n}\n' % (
                  alt(meta.multiplier), alt(meta.offset), alt(meta.alpha)))
    open(os.path.join(HERE, "meta_u8_pretty.json"), "w").write(pretty)

    # a degenerate interval (all values equal -> alpha = 0, offset = -0.0 when the value is -0.0) and integer-valued floats
    data0 = np.full((3, 16), np.float32(-0.0), dtype=np.float32)
    rows0, meta0 = qo.u8_encode(data0, qo.DOT, True)
    open(os.path.join(HERE, "rows_u8_zero.bin"), "wb").write(rows0.tobytes())
    js0 = '{"actual_dim":16,"alpha":%s,"offset":%s,"multiplier":%s,"vector_parameters":%s}' % (
        ryu_f32(meta0.alpha), ryu_f32(meta0.offset), ryu_f32(meta0.multiplier), vp_json(16, 3, "Dot", True))
    open(os.path.join(HERE, "meta_u8_zero.json"), "w").write(js0)

    # PQ: 9 x 6, chunk 4 (vector_division [0,4) [4,6)), given centroids with values across ryu's layouts
    cen = ((rng.random((256, 6), dtype=np.float32) - np.float32(0.5)) *
           np.float32(10.0) ** rng.integers(-9, 9, size=(256, 6)).astype(np.float32)).astype(np.float32)
    cen[0, :4] = [np.float32(1e-7), np.float32(-0.0), np.float32(1.0), np.float32(16777216.0)]
    cen[1, :3] = [np.float32(1.17549435e-38), np.float32(3.4028235e38), np.float32(-1.4e-45)]
    pdata = rng.random((9, 6), dtype=np.float32)
    prow = qo.pq_encode(pdata, 4, cen)
    open(os.path.join(HERE, "rows_pq.bin"), "wb").write(prow.tobytes())
    np.save(os.path.join(HERE, "centroids_pq.npy"), cen)
    cj = "[" + ",".join("[" + ",".join(ryu_f32(x) for x in row) + "]" for row in cen) + "]"
    js = '{"centroids":%s,"vector_division":[{"start":0,"end":4},{"start":4,"end":6}],"vector_parameters":%s}' % (
        cj, vp_json(6, 9, "Dot", False))
    open(os.path.join(HERE, "meta_pq.json"), "w").write(js)

    # binary: 5 x 70 (u8 store), L1 inverted
    bdata = np.where(rng.random((5, 70)) < 0.5, -1.0, 1.0).astype(np.float32)
    brows = qo.bin_encode(bdata)
    open(os.path.join(HERE, "rows_bin.bin"), "wb").write(brows.tobytes())
    open(os.path.join(HERE, "meta_bin.json"), "w").write('{"vector_parameters":%s}' % vp_json(70, 5, "L1", True))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
