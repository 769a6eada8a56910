#!/usr/bin/env python3
"""LDS / barrier audit of one kernel in a hipcc -S listing: per loop nest level, every LDS write (ds_write*,
LDS-DMA), LDS atomic, barrier and LDS read, so that "which wave may read an LDS buffer while another is rewriting it" can be
answered from the listing: a loop that contains LDS reads but neither LDS writes nor barriers reads constant LDS.
   python tools/isa_lds_audit.py listing.s <substring of the kernel's mangled name>"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if key in l and l.split(";")[0].strip().endswith(":") and not l.startswith((".", ";", "\t")))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    # basic blocks with their loop depth (hipcc annotates loop headers and members with "Depth=N")
    blocks, cur = [], {"label": "entry", "depth": 0, "ins": []}
    for l in body[1:]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur)
            d = re.search(r"Depth=(\d+)", l)
            cur = {"label": m.group(1), "depth": int(d.group(1)) if d else 0, "ins": []}
            continue
        d = re.search(r";\s+(?:=>)?\s*(?:This )?(?:Inner )?Loop Header: Depth=(\d+)", l)
        if d and not cur["ins"]:
            cur["depth"] = max(cur["depth"], int(d.group(1)))
        if t and not t.startswith((";", ".")):
            cur["ins"].append(t.split(";")[0].strip())
    blocks.append(cur)
    kinds = (("LDS write", r"^(ds_write|ds_store|buffer_load.* lds|global_load_lds)"), ("LDS atomic", r"^ds_(add|sub|inc|dec|min|max|and|or|xor|cmpst|wrxchg)"),
             ("barrier", r"^s_barrier"), ("LDS read", r"^ds_read"), ("lgkmcnt wait", r"^s_waitcnt.*lgkmcnt"), ("matrix", r"^v_mfma"))
    print(f"{lines[start].split(':')[0]}: {len(body)} lines, {len(blocks)} basic blocks")
    by_depth = {}
    for b in blocks:
        c = by_depth.setdefault(b["depth"], {k: 0 for k, _ in kinds})
        for ins in b["ins"]:
            for k, pat in kinds:
                if re.match(pat, ins):
                    c[k] += 1
    for d in sorted(by_depth):
        print(f"  loop depth {d}: " + ", ".join(f"{k} {v}" for k, v in by_depth[d].items()))
    print("  blocks inside loops that WRITE LDS or hold a barrier:")
    any_ = False
    for b in blocks:
        if b["depth"] == 0:
            continue
        w = [i for i in b["ins"] if re.match(kinds[0][1], i) or re.match(kinds[1][1], i) or re.match(kinds[2][1], i)]
        if w:
            any_ = True
            print(f"    {b['label']} (depth {b['depth']}, {sum(1 for i in b['ins'] if i.startswith('v_mfma'))} matrix instructions): " + "; ".join(w[:6]) + (" ..." if len(w) > 6 else ""))
    if not any_:
        print("    none")


if __name__ == "__main__":
    main()
