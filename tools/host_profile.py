"""The report of tools/host_profile.sh: program counters sampled by SIGPROF (pcs.txt) -> share of the samples per function.
The binary's own addresses go through addr2line (innermost inlined frame), those inside shared libraries are attributed to the
nearest exported symbol below them (nm -D): for libc that names the neighbourhood (malloc's internals show up under the exported
function in front of them), which is all a 250 Hz profile needs."""
import bisect
import collections
import subprocess
import sys

tmp = sys.argv[1] if len(sys.argv) > 1 else "."
maps = []
for line in open(f"{tmp}/maps.txt"):
    f = line.split()
    if len(f) >= 6 and "x" in f[1]:
        lo, hi = (int(x, 16) for x in f[0].split("-"))
        maps.append((lo, hi, int(f[2], 16), f[5]))
samples = collections.Counter(int(line, 16) for line in open(f"{tmp}/pcs.txt"))
total = sum(samples.values())
per_object = collections.defaultdict(collections.Counter)
for pc, n in samples.items():
    for lo, hi, offset, name in maps:
        if lo <= pc < hi:
            per_object[name][pc if name.endswith("host_profile") else pc - lo + offset] += n
            break
    else:
        per_object["(kernel / unknown)"][pc] += n
functions = collections.Counter()
for name, pcs in per_object.items():
    if name.endswith("host_profile"):
        out = subprocess.run(["addr2line", "-f", "-C", "-e", name] + [hex(pc) for pc in pcs], capture_output=True, text=True).stdout.splitlines()
        for (pc, n), i in zip(pcs.items(), range(0, len(out), 2)):
            functions[out[i][:100]] += n
    elif name.startswith("/"):
        symbols = sorted((int(f[0], 16), f[2]) for f in (l.split() for l in subprocess.run(["nm", "-D", "--defined-only", name], capture_output=True, text=True).stdout.splitlines()) if len(f) == 3)
        addresses = [a for a, _ in symbols]
        short = name.rsplit("/", 1)[-1]
        for offset, n in pcs.items():
            i = bisect.bisect_right(addresses, offset) - 1
            functions[f"[{short}] near {symbols[i][1] if i >= 0 else '?'}"] += n
    else:
        functions[name] += sum(pcs.values())
print(f"{total} samples")
for name, n in functions.most_common(40):
    print(f"{100.0 * n / total:5.1f} %  {name}")
