#!/usr/bin/env python3
"""Sum the `[flx host profile]` lines (FLX_HOST_PROFILE=1, stderr of any run) per phase: wall and thread-CPU milliseconds.
usage: host_profile_summary.py LOG [SKIP]     (SKIP: leave out the first SKIP lines of every kind: the warm-up's allocations)"""
import collections
import re
import sys

wall = collections.defaultdict(float)
cpu = collections.defaultdict(float)
count = collections.Counter()
seen = collections.Counter()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for line in open(sys.argv[1], errors="replace"):
    m = re.match(r"\[flx host profile\] (\S+) total \S+ ms:(.*)", line)
    if not m:
        continue
    what = m.group(1)
    seen[what] += 1
    if seen[what] <= skip:
        continue
    count[what] += 1
    for name, w, c in re.findall(r" (\S+)=([\d.]+)/([\d.]+)", m.group(2)):
        wall[(what, name)] += float(w)
        cpu[(what, name)] += float(c)
for what in count:
    tw = sum(v for (a, _), v in wall.items() if a == what)
    tc = sum(v for (a, _), v in cpu.items() if a == what)
    print(f"{what}: {count[what]} calls, wall {tw:.0f} ms, cpu {tc:.0f} ms ({tc / count[what]:.2f} ms cpu per call)")
    for (a, name), v in wall.items():
        if a == what:
            print(f"    {name:20s} wall {v:10.1f}  cpu {cpu[(a, name)]:10.1f}  ({cpu[(a, name)] / count[what]:.2f} per call)")
