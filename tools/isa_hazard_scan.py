"""Static audit of a kernel's gfx950 ISA for register dependences that sit closer than a given number of issue slots:
   * transcendental (v_exp/v_log/v_rcp/v_rsq/v_sqrt) -> reader / overwriter of its destination / overwriter of its SOURCE,
   * matrix instruction -> VALU / LDS / VMEM write of its A, B or C operand registers, reader or writer of its result,
   * VALU write -> matrix instruction reading that register as an operand.
Distances count instructions (s_nop N as N + 1).  Used on the packed-fp32 Gaussian shared-rig kernel that gave launch-to-launch
differences in round 2 (git show 9cc70cf:facedeform_amd/csrc/fd_eval_shared.hip) and on the shipped kernels.
    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only ... -o k.s
    python tools/isa_hazard_scan.py k.s <first line> <last line> [window]"""
import re, sys

TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_f64", "v_rsq_f64")


def regs(op):
    op = op.strip()
    m = re.fullmatch(r"-?\|?([va])(\d+)\|?", op)
    if m:
        return {(m.group(1), int(m.group(2)))}
    m = re.fullmatch(r"-?\|?([va])\[(\d+):(\d+)\]\|?", op)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    return set()


def parse(t):
    parts = t.split(None, 1)
    ops = [o.strip().split(" ")[0] for o in re.split(r",\s*", parts[1])] if len(parts) > 1 else []
    return parts[0], ops


def main():
    fn, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    window = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    ins = []
    for ln in open(fn).read().split("\n")[lo - 1:hi - 1]:
        t = ln.split(";")[0].strip()
        if t and not t.endswith(":") and not t.startswith("."):
            ins.append(t)
    P = [parse(t) for t in ins]
    base = lambda o: o.replace("_e32", "").replace("_e64", "")
    stats = {}

    def walk(i, step):
        k, j = 0, i
        while k < window and 0 <= j + step < len(P):
            j += step
            o2, _ = P[j]
            if o2.startswith("s_nop"):
                k += int(re.search(r"\d+", ins[j]).group()) + 1
                continue
            k += 1
            yield k, j

    def writes(o, p):
        if o.startswith("v_cmp") or o.startswith("v_readlane") or o.startswith("v_readfirstlane"):
            return set()
        if o.startswith(("v_", "ds_read", "global_load", "buffer_load", "scratch_load")):
            return regs(p[0]) if p else set()
        return set()

    def reads(o, p):
        if o.startswith(("v_", "ds_read", "global_load", "buffer_load", "scratch_load")) and not o.startswith("v_cmp"):
            return set().union(*[regs(x) for x in p[1:]]) if len(p) > 1 else set()
        return set().union(*[regs(x) for x in p]) if p else set()

    for i, (opc, ops) in enumerate(P):
        if base(opc) in TRANS:
            d, s = regs(ops[0]), set().union(*[regs(x) for x in ops[1:]])
            for k, j in walk(i, +1):
                o2, p2 = P[j]
                w, r = writes(o2, p2), reads(o2, p2)
                for kind, hit in (("trans source overwritten (WAR)", w & s), ("trans result overwritten (WAW)", w & d), ("trans result read (RAW)", r & d)):
                    if hit:
                        stats.setdefault((kind, k, base(o2).split(" ")[0]), []).append((ins[i], ins[j]))
        if "mfma" in opc:
            D, A, B, C = [regs(x) for x in ops[:4]]
            for k, j in walk(i, +1):
                o2, p2 = P[j]
                if "mfma" in o2:
                    continue
                w, r = writes(o2, p2), reads(o2, p2)
                for kind, hit in (("matrix operand A overwritten", w & A), ("matrix operand B overwritten", w & B), ("matrix operand C overwritten", w & (C - D)),
                                  ("matrix result overwritten", w & D), ("matrix result read", r & D)):
                    if hit:
                        stats.setdefault((kind, k, base(o2)), []).append((ins[i], ins[j]))
            for k, j in walk(i, -1):
                o2, p2 = P[j]
                if not o2.startswith("v_") or "mfma" in o2:
                    continue
                if writes(o2, p2) & (A | B | C):
                    stats.setdefault(("VALU write -> matrix operand read", k, base(o2)), []).append((ins[j], ins[i]))
    for key in sorted(stats):
        ex = stats[key][0]
        print(f"{key[0]:34s} distance {key[1]:2d}  by {key[2]:22s} x{len(stats[key]):4d}   e.g. {ex[0]}  ->  {ex[1]}")


if __name__ == "__main__":
    main()
