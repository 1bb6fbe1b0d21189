"""The fixed BLAKE3 instruction order (zinc_amd/csrc/blake3_sched.inc) is GENERATED: the generator's own interpreter must
reproduce a plain BLAKE3 compression, the published BLAKE3("") / BLAKE3("abc") digests and the vectors the oracle is pinned
with, and the committed file must be what the generator emits today."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_blake3_sched", os.path.join(ROOT, "tools", "gen_blake3_sched.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_schedule_computes_blake3():
    g = _gen()
    g.check()  # 200 random messages per body against the reference compression + the two published digests


def test_schedule_against_the_golden_blake3_vectors():
    """One-block messages of tests/golden/blake3_vectors.json (generated from the official C implementation) through the
    SCHEDULED instruction lists: 64-byte messages through the NODE body, 32-byte ones through the HALF body."""
    g = _gen()
    with open(os.path.join(ROOT, "tests", "golden", "blake3_vectors.json")) as fh:
        vec = json.load(fh)
    cases = [v for v in (vec["vectors"] if isinstance(vec, dict) and "vectors" in vec else vec) if isinstance(v, dict)]
    node, half = g.schedule(g.build(16, 64)), g.schedule(g.build(8, 32))
    done = 0
    for v in cases:
        msg = bytes.fromhex(v["msg"]) if "msg" in v else None
        want = v.get("hash_hex") or v.get("digest_hex") or v.get("hash")
        if msg is None or want is None or len(msg) not in (32, 64):
            continue
        words = [int.from_bytes(msg[4 * i:4 * i + 4], "little") for i in range(len(msg) // 4)]
        got = g.interpret(node if len(msg) == 64 else half, words)
        assert b"".join(w.to_bytes(4, "little") for w in got).hex() == want[:64], len(msg)
        done += 1
    assert done >= 2, "the golden file holds no 32- or 64-byte single-block vectors any more?"


def test_committed_include_is_current():
    """zinc_amd/csrc/blake3_sched.inc == what tools/gen_blake3_sched.py renders (the generator is deterministic).  In
    process: no fork + exec (a GPU-initialised pytest process must not), no write to the tracked file."""
    g = _gen()
    with open(g.INC_PATH) as fh:
        assert fh.read() == g.render(), "blake3_sched.inc is stale: run tools/gen_blake3_sched.py and rebuild"
