import sys, collections
d = collections.defaultdict(list)
for line in open(sys.argv[1]):
    p = line.split()
    if len(p) == 3: d[p[0]].append(float(p[2]))
for k, v in d.items():
    print("%-20s n %2d mean %.3f M  min %.3f max %.3f" % (k, len(v), sum(v)/len(v)/1e6, min(v)/1e6, max(v)/1e6))
