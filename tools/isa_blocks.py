"""per-basic-block instruction mix of kernels in a hipcc -S listing: python tools/isa_blocks.py file.s substr [substr ...]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
cur_fn, blocks, cur = None, [], None
def flush():
    if cur_fn and any(w in cur_fn for w in want):
        print(cur_fn[:100])
        for name, ins in blocks + ([cur] if cur else []):
            if len(ins) < 16:
                continue
            c = collections.Counter()
            for i in ins:
                if i.startswith("v_"):
                    c["v_f64" if "f64" in i and not i.startswith("v_cvt") else ("v_cvt" if i.startswith("v_cvt") else "v_other")] += 1
                elif i.startswith("ds_"):
                    c["ds"] += 1
                elif i.startswith("s_waitcnt"):
                    c["wait"] += 1
                elif i.startswith("s_"):
                    c["s"] += 1
                elif i.startswith(("global_", "buffer_", "flat_")):
                    c["vmem"] += 1
                else:
                    c["other"] += 1
            print("   %-12s %4d  %s" % (name, len(ins), dict(sorted(c.items()))))
for line in txt:
    m = re.match(r"^(_Z\S+):", line)
    if m:
        flush()
        cur_fn, blocks, cur = m.group(1), [], ("entry", [])
        continue
    ls = line.strip()
    if cur_fn is None or not ls:
        continue
    if ls.startswith(".Lfunc_end"):
        flush()
        cur_fn = None
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", ls)
    if m:
        blocks.append(cur)
        cur = (m.group(1), [])
    elif not ls.startswith((";", ".")):
        cur[1].append(ls.split()[0])
