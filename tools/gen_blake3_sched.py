#!/usr/bin/env python3
"""Generates zinc_amd/csrc/blake3_sched.inc: one BLAKE3 compression (cv = IV, counter 0, flags CHUNK_START | CHUNK_END |
ROOT) as a FIXED instruction order for gfx950.

Why: on gfx950 the two-operand VOP2 integer opcodes (v_xor_b32, v_add_u32) issue at twice the rate of the VOP3 ones
(v_add3_u32, v_alignbit_b32) -- 2.7 against 4.1 cycles per wave64 instruction per SIMD, tools/ubench_valu_ops -- but a
SIMD only gets that when a wave's stream ALTERNATES the two kinds: the four independent G functions of a half-round
issued S F S F ... run at 3.37 cycles per instruction, the order hipcc's scheduler picks at 3.98, one G after the other
at 4.33 (tools/ubench_gsched, profiles/round3_valu_issue.md).  The compiler cannot be told about opcode classes, so
the order is fixed here: every instruction is its own `asm volatile` statement (volatile asm statements keep their
order; register allocation stays with the compiler).

Two bodies:
  NODE  16 message words in VGPRs, block_len 64     (tree nodes: src/zip/pcs/utils.rs:107-112)
  HALF  8 message words in VGPRs, words 8..15 zero, block_len 32   (leaves of Int<4>: src/field/int.rs:201-210)
The initial state is constant (IV, counter, block_len, flags): constants are folded as hipcc folds them (a VOP2 with a
32-bit literal), so the instruction count equals the compiler's.

`python3 tools/gen_blake3_sched.py --check` interprets the generated order on random messages against a plain Python
BLAKE3 compression (and the BLAKE3("") / "abc" vectors) before anything reaches a GPU.
"""
import argparse
import os
import random

IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
PERM = [2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8]
FLAGS = 0x0B
M32 = 0xFFFFFFFF


def rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M32


def ref_compress(m, block_len):
    v = IV + IV[:4] + [0, 0, block_len, FLAGS]
    m = list(m)

    def g(a, b, c, d, mx, my):
        v[a] = (v[a] + v[b] + mx) & M32; v[d] = rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & M32; v[b] = rotr(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + my) & M32; v[d] = rotr(v[d] ^ v[a], 8)
        v[c] = (v[c] + v[d]) & M32; v[b] = rotr(v[b] ^ v[c], 7)

    for r in range(7):
        g(0, 4, 8, 12, m[0], m[1]); g(1, 5, 9, 13, m[2], m[3]); g(2, 6, 10, 14, m[4], m[5]); g(3, 7, 11, 15, m[6], m[7])
        g(0, 5, 10, 15, m[8], m[9]); g(1, 6, 11, 12, m[10], m[11]); g(2, 7, 8, 13, m[12], m[13]); g(3, 4, 9, 14, m[14], m[15])
        m = [m[p] for p in PERM]
    return [v[i] ^ v[i + 8] for i in range(8)]


class Builder:
    """Canonical-order op list with constant folding.  Operands: ('c', value) | ('r', name)."""

    def __init__(self, n_msg):
        self.ops = []  # (kind, dst, srcs, cls): kind in add3 | add | xor | rot ; srcs operands ; rot carries n
        self.val = {}  # name -> ('c', K) for names that are still compile-time constants
        self.n_msg = n_msg

    def get(self, name):
        return self.val.get(name, ('r', name))

    def msg(self, i):
        return ('r', f"m{i}") if i < self.n_msg else ('c', 0)

    def emit(self, kind, dst, srcs, extra=None):
        cls = 'S' if kind in ('add3', 'rot') else 'F'
        self.ops.append({'kind': kind, 'dst': dst, 'srcs': srcs, 'n': extra, 'cls': cls})
        self.val.pop(dst, None)

    def add_n(self, dst, terms):
        k = sum(t[1] for t in terms if t[0] == 'c') & M32
        regs = [t for t in terms if t[0] == 'r']
        if not regs:
            self.val[dst] = ('c', k)
        elif len(regs) == 1 and k == 0:
            if regs[0][1] != dst:
                raise AssertionError("plain move not expected")
        elif len(regs) == 1:
            self.emit('add', dst, [('c', k), regs[0]])
        elif len(regs) == 2 and k == 0:
            self.emit('add', dst, regs)
        elif len(regs) == 2:
            raise AssertionError("reg + reg + constant does not occur in BLAKE3 with a constant initial state")
        else:
            self.emit('add3', dst, regs)

    def xor_rot(self, dst, a, b, n):
        """dst = rotr(a ^ b, n)"""
        if a[0] == 'c' and b[0] == 'c':
            self.val[dst] = ('c', rotr(a[1] ^ b[1], n))
            return
        if a[0] == 'c' or b[0] == 'c':
            c, r = (a, b) if a[0] == 'c' else (b, a)
            if c[1] == 0:
                self.emit('rot', dst, [r], n)
                return
            self.emit('xor', dst, [c, r])
        else:
            self.emit('xor', dst, [a, b])
        self.emit('rot', dst, [('r', dst)], n)

    def g(self, a, b, c, d, mx, my):
        A, B, C, D = f"v{a}", f"v{b}", f"v{c}", f"v{d}"
        self.add_n(A, [self.get(A), self.get(B), mx])
        self.xor_rot(D, self.get(D), self.get(A), 16)
        self.add_n(C, [self.get(C), self.get(D)])
        self.xor_rot(B, self.get(B), self.get(C), 12)
        self.add_n(A, [self.get(A), self.get(B), my])
        self.xor_rot(D, self.get(D), self.get(A), 8)
        self.add_n(C, [self.get(C), self.get(D)])
        self.xor_rot(B, self.get(B), self.get(C), 7)


def build(n_msg, block_len):
    b = Builder(n_msg)
    init = IV + IV[:4] + [0, 0, block_len, FLAGS]
    for i, k in enumerate(init):
        b.val[f"v{i}"] = ('c', k)
    idx = list(range(16))
    for _ in range(7):
        m = [b.msg(i) for i in idx]
        b.g(0, 4, 8, 12, m[0], m[1]); b.g(1, 5, 9, 13, m[2], m[3]); b.g(2, 6, 10, 14, m[4], m[5]); b.g(3, 7, 11, 15, m[6], m[7])
        b.g(0, 5, 10, 15, m[8], m[9]); b.g(1, 6, 11, 12, m[10], m[11]); b.g(2, 7, 8, 13, m[12], m[13]); b.g(3, 4, 9, 14, m[14], m[15])
        idx = [idx[p] for p in PERM]
    for i in range(8):
        assert f"v{i}" not in b.val and f"v{i + 8}" not in b.val
        b.emit('xor', f"h{i}", [('r', f"v{i}"), ('r', f"v{i + 8}")])
    return b.ops


def schedule(ops):
    """List scheduling over the RAW / WAR / WAW edges of the canonical order; objective: alternate the classes, among
    the candidates take the one with the longest chain of dependants behind it."""
    n = len(ops)
    succ = [set() for _ in range(n)]
    npred = [0] * n
    last_write, readers = {}, {}
    for i, op in enumerate(ops):
        deps = set()
        for s in op['srcs']:
            if s[0] == 'r' and s[1] in last_write:
                deps.add(last_write[s[1]])
        d = op['dst']
        if d in last_write:
            deps.add(last_write[d])
        for r in readers.get(d, ()):
            if r != i:
                deps.add(r)
        for p in deps:
            if i not in succ[p]:
                succ[p].add(i)
                npred[i] += 1
        for s in op['srcs']:
            if s[0] == 'r':
                readers.setdefault(s[1], []).append(i)
        last_write[d] = i
        readers[d] = []
    height = [0] * n
    for i in range(n - 1, -1, -1):
        height[i] = 1 + max((height[j] for j in succ[i]), default=0)
    ready = [i for i in range(n) if npred[i] == 0]
    order, want = [], 'S'
    while ready:
        cand = [i for i in ready if ops[i]['cls'] == want] or ready
        pick = max(cand, key=lambda i: (height[i], -i))
        ready.remove(pick)
        order.append(pick)
        want = 'F' if ops[pick]['cls'] == 'S' else 'S'
        for j in succ[pick]:
            npred[j] -= 1
            if npred[j] == 0:
                ready.append(j)
    assert len(order) == n
    return [ops[i] for i in order]


def interpret(ops, m):
    env = {f"m{i}": w for i, w in enumerate(m)}

    def val(o):
        return o[1] if o[0] == 'c' else env[o[1]]

    for op in ops:
        s = [val(x) for x in op['srcs']]
        if op['kind'] in ('add', 'add3'):
            env[op['dst']] = sum(s) & M32
        elif op['kind'] == 'xor':
            env[op['dst']] = s[0] ^ s[1]
        else:
            env[op['dst']] = rotr(s[0], op['n'])
    return [env[f"h{i}"] for i in range(8)]


def check():
    rng = random.Random(1)
    for n_msg, bl in ((16, 64), (8, 32)):
        ops = schedule(build(n_msg, bl))
        for _ in range(200):
            m = [rng.getrandbits(32) for _ in range(n_msg)] + [0] * (16 - n_msg)
            assert interpret(ops, m[:n_msg]) == ref_compress(m, bl)
        cls = "".join(o['cls'] for o in ops)
        print(f"n_msg {n_msg}: {len(ops)} instructions, {cls.count('S')} single-rate / {cls.count('F')} double-rate, "
              f"{cls.count('SS')} SS pairs, {cls.count('FF')} FF pairs")
    # published vectors: BLAKE3("") and BLAKE3("abc") are one compression each (block_len 0 / 3: through the reference
    # function only -- pins ref_compress itself)
    def digest(words):
        return b"".join(w.to_bytes(4, "little") for w in words).hex()
    assert digest(ref_compress([0] * 16, 0)) == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert digest(ref_compress([int.from_bytes(b"abc\0", "little")] + [0] * 15, 3)) == \
        "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"
    print("check ok")


def operand(o, is_dst_named):
    raise NotImplementedError


def emit_cpp(name, ops, n_msg):
    lines = [f"__device__ __forceinline__ void {name}(const uint32_t (&m)[{n_msg}], uint32_t (&h)[8]) {{",
             "    uint32_t " + ", ".join(f"v{i}" for i in range(16)) + ";"]
    defined = set()

    def cexpr(nm):
        if nm.startswith("m"):
            return f"m[{nm[1:]}]"
        if nm.startswith("h"):
            return f"h[{nm[1:]}]"
        return nm

    for op in ops:
        dst = op['dst']
        srcs = op['srcs']
        reads_dst = any(s == ('r', dst) for s in srcs)
        outs, ins, args = [], [], []

        def reg(nm):
            if nm == dst:
                return "%0"
            key = cexpr(nm)
            if key not in ins:
                ins.append(key)
            return f"%{1 + ins.index(key)}"

        def lit(k):
            return f"0x{k:x}"

        if op['kind'] == 'add3':
            t = f"v_add3_u32 %0, {reg(srcs[0][1])}, {reg(srcs[1][1])}, {reg(srcs[2][1])}"
        elif op['kind'] in ('add', 'xor'):
            mnem = "v_add_u32" if op['kind'] == 'add' else "v_xor_b32"
            a, b = srcs
            if a[0] == 'c':
                t = f"{mnem} %0, {lit(a[1])}, {reg(b[1])}"
            else:
                t = f"{mnem} %0, {reg(a[1])}, {reg(b[1])}"
        else:
            r = reg(srcs[0][1])
            t = f"v_alignbit_b32 %0, {r}, {r}, {op['n']}"
        con = ('"+v"' if reads_dst else '"=v"') + f"({cexpr(dst)})"
        assert reads_dst or dst not in defined or True
        defined.add(dst)
        lines.append(f'    asm volatile("{t}" : {con} : ' + ", ".join(f'"v"({x})' for x in ins) + ");")
    # (the next instruction the compiler emits may read h[] through DPP: two wait states after a VALU write)
    lines.append('    asm volatile("s_nop 1");')
    lines.append("}")
    return lines


INC_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zinc_amd", "csrc", "blake3_sched.inc")


def render():
    """The text of zinc_amd/csrc/blake3_sched.inc (deterministic)."""
    out = ["// GENERATED by tools/gen_blake3_sched.py -- do not edit (see that file for the why and the how).",
           "// One BLAKE3 compression (cv = IV, counter 0, flags 0x0B) in a fixed, class-alternating instruction order.",
           "#pragma once", "#include <stdint.h>", "", "namespace zipk {", ""]
    node = schedule(build(16, 64))
    half = schedule(build(8, 32))
    out.append("// 64-byte message (tree node): " + "".join(o['cls'] for o in node)[:96] + "...")
    out += emit_cpp("blake3_sched_node", node, 16)
    out.append("")
    out.append("// 32-byte message, words 8..15 zero (Int<4> leaf): " + "".join(o['cls'] for o in half)[:96] + "...")
    out += emit_cpp("blake3_sched_half", half, 8)
    out += ["", "}  // namespace zipk", ""]
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    check()
    if args.check:
        return
    with open(INC_PATH, "w") as fh:
        fh.write(render())
    print("wrote", INC_PATH)


if __name__ == "__main__":
    main()
