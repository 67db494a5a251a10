#!/usr/bin/env python3
"""Regenerates tests/golden/ from the reference checkout (run in the build container only).

Fixtures are DATA the reference's own tests hold, no reference source text:
  * decodecorpus/zNNNNNN.zst        -- the 100 compressed golden inputs
    (reference: decodecorpus_files/, driven by cmd/sparkzstd/main.go:66-108)
  * decodecorpus/zNNNNNN            -- expected output, verbatim when <= 64 KiB
  * decodecorpus/manifest.json      -- length + sha256 of every expected output
  * kat.json                        -- known-answer vectors transcribed as data:
      - predefined literal-length FSE decode table (fse/fse_test.go:8-41)
      - reverse bitstream edge reads (bitstream/reversebitstream_test.go:172-229)
      - ring buffer string KATs (decompression/ringbuffer_test.go:9-154)
"""
import hashlib, json, os, re, shutil, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = 64 * 1024


def corpus():
    out = os.path.join(HERE, "decodecorpus")
    os.makedirs(out, exist_ok=True)
    manifest = {}
    src = os.path.join(REF, "decodecorpus_files")
    for name in sorted(os.listdir(src)):
        if not name.endswith(".zst"):
            continue
        stem = name[:-4]
        shutil.copyfile(os.path.join(src, name), os.path.join(out, name))
        data = open(os.path.join(src, stem), "rb").read()
        manifest[stem] = {"length": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                          "verbatim": len(data) <= SMALL}
        dst = os.path.join(out, stem)
        if len(data) <= SMALL:
            open(dst, "wb").write(data)
        elif os.path.exists(dst):
            os.remove(dst)
    json.dump(manifest, open(os.path.join(out, "manifest.json"), "w"), indent=1, sort_keys=True)


def kats():
    kat = {}
    # fse_test.go:8-41: {Baseline, NumberOfAdditionalBits, NumberOfBits, Symbol(base value)}
    txt = open(os.path.join(REF, "fse", "fse_test.go")).read()
    body = txt[txt.index("expectedLLDecodingTable"):txt.index("func TestBuilding")]
    rows = re.findall(r"\{(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\}", body)
    assert len(rows) == 64
    kat["ll_predefined_table"] = [[int(x) for x in r] for r in rows]
    # reversebitstream_test.go:172-229 TestEdges
    kat["rbs_edges"] = {"data": [64, 58, 169, 224],
                        "reads": [3, 4, 4, 1, 3, 5, 1, 3, 3, 0, 4],
                        "expect": [7, 0, 5, 0, 4, 19, 1, 2, 2, 0, 0]}
    # ringbuffer_test.go:9-83 TestRingbuffer: (op, arg, window string after, dumped by this op)
    kat["ring_push"] = {"len": 10, "steps": [
        ["push", "Teststring", "Teststring", ""],
        ["push", "AABB", "stringAABB", "Test"],
        ["push", "123456789012345678", "9012345678", "stringAABB12345678"],
        ["push", "ABCDEFGH", "78ABCDEFGH", "90123456"]]}
    # ringbuffer_test.go:85-154 TestRepeat
    kat["ring_repeat"] = {"len": 10, "steps": [
        ["push", "Teststring", "Teststring", ""],
        ["repeat", [4, 0], "stringring", "Test"],
        ["repeat", [8, 0], "ngringring", "stringri"],
        ["push", "1234567890", "1234567890", "ngringring"],
        ["repeat", [5, 3], "6789034567", "12345"],
        ["repeat", [3, 7], "9034567678", "678"]]}
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    corpus()
    kats()
    print("golden fixtures written to", HERE)
