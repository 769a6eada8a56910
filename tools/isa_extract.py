#!/usr/bin/env python3
"""Cut one kernel's body out of a hipcc -S listing and count its instructions.
   python tools/isa_extract.py listing.s <substring of the mangled name> [out.s]"""
import collections
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if key in l and l.split(";")[0].strip().endswith(":") and not l.startswith((".", ";", "\t")):
            start = i
            break
    if start is None:
        raise SystemExit(f"no label containing {key}")
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write("\n".join(body))
    c = collections.Counter()
    for l in body:
        t = l.strip().split(" ")[0]
        if t and not t.startswith((".", ";", "_")) and not t.endswith(":"):
            c[t] += 1
    print(lines[start], len(body), "lines")
    for k, v in c.most_common(60):
        print(f"  {k:32s} {v}")


if __name__ == "__main__":
    main()
