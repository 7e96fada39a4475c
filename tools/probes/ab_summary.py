import sys, collections
r = collections.defaultdict(list)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) > 6 and p[2].replace(".", "").isdigit():
        r[p[0]].append((float(p[4]), float(p[-1])))
for k, v in r.items():
    it = sorted(x[0] for x in v); ke = sorted(x[1] for x in v)
    print(f"{k:16s} n {len(v)}  iteration ms: mean {sum(it)/len(it):.4f} median {it[len(it)//2]:.4f} min {it[0]:.4f}   kernel ms: mean {sum(ke)/len(ke):.4f} median {ke[len(ke)//2]:.4f} min {ke[0]:.4f}")
