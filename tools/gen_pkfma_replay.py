"""dev: turn tools/pkfma_tail_block.s (the verbatim tail of round 3's faulty build) into inline asm for tools/pkfma_probe.hip:
   PK_REPLAY_FAST  = input set-up + the block exactly as the compiler scheduled it
   PK_REPLAY_SAFE  = the same set-up + the same instructions, each followed by s_nop 7 x 2 and a full s_waitcnt (no hazard can survive)
   PK_REPLAY_CLOBBERS = every register the block touches
 writes tools/pkfma_replay.inc"""
import os, re, struct
here = os.path.dirname(os.path.abspath(__file__))
blk = [l.strip() for l in open(os.path.join(here, "pkfma_tail_block.s")) if l.strip() and not l.strip().startswith(";")]
regs, written = set(), set()
for l in blk:
    ops = l.split(None, 1)[1] if " " in l else ""
    toks = re.findall(r"v\[(\d+):(\d+)\]|v(\d+)", ops)
    for a, b, c in toks:
        rs = range(int(a), int(b) + 1) if a else [int(c)]
        regs.update(rs)
# inputs = registers read before they are written
inputs, seen_w = [], set()
for l in blk:
    name, ops = l.split(None, 1) if " " in l else (l, "")
    parts = [p.strip() for p in ops.split(",")]
    def expand(p):
        m = re.match(r"v\[(\d+):(\d+)\]", p)
        if m: return list(range(int(m.group(1)), int(m.group(2)) + 1))
        m = re.match(r"v(\d+)\b", p)
        return [int(m.group(1))] if m else []
    if name.startswith("global_store") or name.startswith("s_"):
        srcs, dst = sum((expand(p) for p in parts), []), []
    else:
        dst, srcs = expand(parts[0]), sum((expand(p) for p in parts[1:]), [])
    for r in srcs:
        if r not in seen_w and r not in inputs: inputs.append(r)
    seen_w.update(dst)
special = {103, 101, 162, 163, 180, 209, 211}       # 209 / 211: the unused high halves of the broadcast pairs v[208:209], v[210:211]
data_inputs = [r for r in inputs if r not in special]
assert len(data_inputs) == 128, len(data_inputs)
setup = []
for i, r in enumerate(sorted(data_inputs)):
    f = 0.5 + (i * 37 % 128) / 128.0
    setup.append("v_mul_f32_e32 v%d, 0x%08x, %%[seed]" % (r, struct.unpack("<I", struct.pack("<f", f))[0]))
setup += ["v_mov_b32_e32 v103, 0", "v_mov_b32_e32 v101, 0", "v_mov_b32_e32 v209, 0", "v_mov_b32_e32 v211, 0", "v_mov_b32_e32 v162, %[slot]", "v_mov_b32_e32 v163, 0", "s_mov_b64 s[70:71], %[base]",
          "v_mov_b32_e32 v180, %[lds]", "s_nop 7"]
def q(lines): return " \\\n    ".join('"%s\\n\\t"' % l for l in lines)
def rename(lines, m):          # rename VGPRs (whole numbers and the ends of ranges) by the map m
    def sub_range(mo):
        a, b = int(mo.group(1)), int(mo.group(2))
        return "v[%d:%d]" % (m.get(a, a), m.get(a, a) + (b - a))
    out = []
    for l in lines:
        l = re.sub(r"v\[(\d+):(\d+)\]", sub_range, l)
        l = re.sub(r"\bv(\d+)\b", lambda mo: "v%d" % m.get(int(mo.group(1)), int(mo.group(1))), l)
        out.append(l)
    return out
tail = ["s_nop 7", "s_waitcnt vmcnt(0)"]
safe = setup[:]
for l in blk:
    safe += [l, "s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 7", "s_nop 7"]
safe += ["s_waitcnt vmcnt(0)"]
# variants of the FAST block (what changes, nothing else):
variants = {}
variants[0] = ("verbatim", setup + blk + tail)
# 1: the eight row weights do not come from LDS: v_mov of the same values (from the operands %[w0] .. %[w7]) in place of the two ds_read_b128
nolds = []
for l in blk:
    if l.startswith("ds_read_b128 v[176:179]"): nolds += ["v_mov_b32_e32 v%d, %%[w%d]" % (176 + i, i) for i in range(4)]
    elif l.startswith("ds_read_b128 v[180:183]"): nolds += ["v_mov_b32_e32 v%d, %%[w%d]" % (180 + i, 4 + i) for i in range(4)]
    else: nolds.append(l)
variants[1] = ("row weights by v_mov instead of ds_read_b128", setup + nolds + tail)
# 2: everything the LDS returns has landed (and then some) before the first instruction that touches those registers
drained = []
for l in blk:
    drained.append(l)
    if l.startswith("ds_read_b128 v[180:183]"): drained += ["s_waitcnt lgkmcnt(0)", "s_nop 7", "s_nop 7"]
variants[2] = ("s_waitcnt lgkmcnt(0) + 16 wait states behind the two ds_read_b128", setup + drained + tail)
# 3: the accumulators do not reuse the ds_read destinations: every VALU write of v176..v179 (and its later reads) goes to v212..v215
acc, seen = [], False
for l in blk:
    name = l.split()[0]
    if name.startswith("v_pk_fma") and re.match(r"v_pk_fma_f32 v\[17[68]:17[79]\]", l): seen = True
    if not seen or name.startswith("ds_read"):
        acc.append(l)
        continue
    # from the first accumulator write on: v176..v179 as DESTINATION or src2 accumulator are the accumulators; as src1 (weights) they are not.
    parts = [x.strip() for x in l.split(None, 1)[1].split(",")]
    if name.startswith("v_pk_fma"):
        ren = lambda x: rename([x], {176: 212, 177: 213, 178: 214, 179: 215})[0]
        d, a0, a1, a2 = parts[0], parts[1], parts[2], ",".join(parts[3:])
        wrote = getattr(rename, "wrote", set())
        if re.match(r"v\[17[68]:", d):
            wrote.add(d); d = ren(d)
        if re.match(r"v\[17[68]:", a2.split()[0]) and a2.split()[0] in wrote:
            a2 = ren(a2.split()[0]) + a2[len(a2.split()[0]):]
        rename.wrote = wrote
        acc.append("%s %s, %s, %s, %s" % (name, d, a0, a1, a2))
    elif name.startswith("global_store") and "v[176:179]" in l:
        acc.append(l.replace("v[176:179]", "v[212:215]"))
    else:
        acc.append(l)
variants[3] = ("accumulators in v212..v215 instead of the ds_read destinations v176..v179", setup + acc + tail)
# 4: one idle issue slot between any two instructions
spaced = []
for l in blk: spaced += [l, "s_nop 0"]
variants[4] = ("s_nop 0 between all instructions", setup + spaced + tail)
# ---- round 2 of the bisection: the reference is the verbatim block (stable across every run), the streams under test are PADDED ones
PAD = ["s_waitcnt vmcnt(0) lgkmcnt(0)", "s_nop 7", "s_nop 7"]
def padded(lines, when=lambda i, l: True, pad=PAD):
    out = []
    for i, l in enumerate(lines):
        out.append(l)
        if when(i, l): out += pad
    return out
ispk = lambda i, l: l.startswith("v_pk_fma")
half = len(blk) // 2
variants[5] = ("PADDED: s_waitcnt + 2 x s_nop 7 behind every instruction", setup + padded(blk) + tail)
variants[6] = ("PADDED behind the v_pk_fma_f32 only", setup + padded(blk, ispk) + tail)
variants[7] = ("PADDED behind everything but the v_pk_fma_f32", setup + padded(blk, lambda i, l: not ispk(i, l)) + tail)
variants[8] = ("PADDED with s_nop 7 only", setup + padded(blk, pad=["s_nop 7"]) + tail)
variants[9] = ("PADDED with s_waitcnt only", setup + padded(blk, pad=["s_waitcnt vmcnt(0) lgkmcnt(0)"]) + tail)
variants[10] = ("PADDED, row weights by v_mov instead of ds_read_b128", setup + padded(nolds) + tail)
variants[11] = ("PADDED, accumulators in v212..v215", setup + padded(acc) + tail)
variants[12] = ("PADDED first half of the block only", setup + padded(blk, lambda i, l: i < half) + tail)
variants[13] = ("PADDED second half of the block only", setup + padded(blk, lambda i, l: i >= half) + tail)
variants[14] = ("PADDED with v_nop x 4 only (VALU idle slots, no scalar instruction)", setup + padded(blk, pad=["v_nop", "v_nop", "v_nop", "v_nop"]) + tail)
# ---- ground truth: symbolic execution of the block -> per output column the ordered list of (input multiplier, row-weight index) terms
lit = {}
for i, r in enumerate(sorted(data_inputs)):
    lit[r] = 0.5 + (i * 37 % 128) / 128.0
sym = {r: ("in", lit[r]) for r in data_inputs}
outputs = []
for l in blk:
    name, ops = (l.split(None, 1) + [""])[:2]
    if name == "ds_read_b128":
        base = int(re.match(r"v\[(\d+)", ops).group(1)); off = 4 if "offset:16" in ops else 0
        for i in range(4): sym[base + i] = ("w", off + i)
    elif name == "v_pk_fma_f32":
        opsel, opselhi = [0, 0, 0], [1, 1, 1]
        for k, a, b, c in re.findall(r"(op_sel(?:_hi)?):\[(\d),(\d),(\d)\]", ops):
            if k == "op_sel": opsel = [int(a), int(b), int(c)]
            else: opselhi = [int(a), int(b), int(c)]
        toks = [t.strip() for t in re.sub(r"op_sel.*", "", ops).split(",") if t.strip()]
        def rd(tok, sel):
            m = re.match(r"v\[(\d+):", tok)
            return sym[int(m.group(1)) + sel] if m else ("acc", [])
        d = int(re.match(r"v\[(\d+)", toks[0]).group(1))
        res = []
        for sel in (opsel, opselhi):
            a, b, c = rd(toks[1], sel[0]), rd(toks[2], sel[1]), rd(toks[3], sel[2])
            assert a[0] == "in" and b[0] == "w" and c[0] == "acc", (l, a, b, c)
            res.append(("acc", c[1] + [(a[1], b[1])]))
        sym[d], sym[d + 1] = res
    elif name == "v_mov_b32_e32":
        d, s_ = [t.strip() for t in ops.split(",")]
        sym[int(d[1:])] = sym.get(int(s_[1:]), ("other",))
    elif name.startswith("global_store"):
        m = re.search(r", v\[(\d+):(\d+)\]", ops)
        outputs += [sym[i] for i in range(int(m.group(1)), int(m.group(2)) + 1)]
assert len(outputs) == 8 and all(o[0] == "acc" and len(o[1]) == 16 for o in outputs)
truth = "static __device__ const float PK_TRUTH_F[8][16] = {" + ", ".join("{" + ", ".join("%.10ef" % t[0] for t in o[1]) + "}" for o in outputs) + "};\n"
truth += "static __device__ const int PK_TRUTH_W[8][16] = {" + ", ".join("{" + ", ".join("%d" % t[1] for t in o[1]) + "}" for o in outputs) + "};\n"
# 15: the control -- every v_pk_fma_f32 as two scalar v_fma_f32 on the same registers (high half through v216: a destination may be a source)
scalar = []
for l in blk:
    if not l.startswith("v_pk_fma_f32"):
        scalar.append(l); continue
    ops = l.split(None, 1)[1]
    opsel, opselhi = [0, 0, 0], [1, 1, 1]
    for k, a, b, c in re.findall(r"(op_sel(?:_hi)?):\[(\d),(\d),(\d)\]", ops):
        if k == "op_sel": opsel = [int(a), int(b), int(c)]
        else: opselhi = [int(a), int(b), int(c)]
    toks = [t.strip() for t in re.sub(r"op_sel.*", "", ops).split(",") if t.strip()]
    def reg(tok, sel):
        m = re.match(r"v\[(\d+):", tok)
        return "v%d" % (int(m.group(1)) + sel) if m else tok
    d = int(re.match(r"v\[(\d+)", toks[0]).group(1))
    scalar.append("v_fma_f32 v216, %s, %s, %s" % (reg(toks[1], opselhi[0]), reg(toks[2], opselhi[1]), reg(toks[3], opselhi[2])))
    scalar.append("v_fma_f32 v%d, %s, %s, %s" % (d, reg(toks[1], opsel[0]), reg(toks[2], opsel[1]), reg(toks[3], opsel[2])))
    scalar.append("v_mov_b32_e32 v%d, v216" % (d + 1))
variants[15] = ("CONTROL: every v_pk_fma_f32 as two v_fma_f32 (same registers, same order)", setup + scalar + tail)
variants[16] = ("CONTROL, PADDED with s_nop 7", setup + padded(scalar, pad=["s_nop 7"]) + tail)
clob = sorted(regs | special | {212, 213, 214, 215, 216})
out = "// generated by tools/gen_pkfma_replay.py from tools/pkfma_tail_block.s -- do not edit\n"
for k, (what, lines) in sorted(variants.items()):
    out += "// variant %d: %s\n#define PK_REPLAY_FAST_%d \\\n    " % (k, what, k) + q(lines) + "\n"
out += "#define PK_REPLAY_NVARIANTS %d\n" % len(variants)
out += "#define PK_REPLAY_NAMES " + ", ".join('"%s"' % v[0] for _, v in sorted(variants.items())) + "\n"
out += truth
out += "#define PK_REPLAY_SAFE \\\n    " + q(safe) + "\n"
out += "#define PK_REPLAY_CLOBBERS " + ", ".join('"v%d"' % r for r in clob) + ', "s70", "s71", "memory"\n'
open(os.path.join(here, "pkfma_replay.inc"), "w").write(out)
print("inputs", len(data_inputs), "registers", len(clob), "instructions", len(blk))
for k, (what, lines) in sorted(variants.items()):
    print(k, what, len(lines))
