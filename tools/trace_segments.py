import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last complete iterate: find last sfm/between linearize kernel as marker
names = [r["Kernel_Name"] for r in rows]
marks = [i for i, n in enumerate(names) if "linearize" in n or "generic_factor_kernel" in n]
# take the segment between the 2 last linearize launches groups
# groups of consecutive linearize launches = starts of LM iterations; segment number argv[3] counted from the end (default 1 = last)
back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
j = marks[-1]
for _ in range(back - 1):
    while j > 0 and ("linearize" in names[j - 1] or "generic_factor_kernel" in names[j - 1]): j -= 1
    j -= 1
    while j > 0 and not ("linearize" in names[j] or "generic_factor_kernel" in names[j]): j -= 1
while j > 0 and ("linearize" in names[j - 1] or "generic_factor_kernel" in names[j - 1]): j -= 1
k = j - 1
while k > 0 and not ("linearize" in names[k] or "generic_factor_kernel" in names[k]): k -= 1
seg = rows[k + 1:j]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = t0
tot_gap = 0; tot_dur = 0
agg = collections.OrderedDict()
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    n = r["Kernel_Name"].split("(")[0][:40]
    a = agg.setdefault(n, [0, 0, 0]); a[0] += 1; a[1] += e - s; a[2] += max(gap, 0)
    tot_gap += max(gap, 0); tot_dur += e - s
    if len(sys.argv) > 2: print(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:7.1f} us  {n}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}")
    prev_end = max(prev_end, e)
print(f"segment: {len(seg)} launches, span {(prev_end - t0) / 1e3:.1f} us, kernel time {tot_dur / 1e3:.1f} us, gaps {tot_gap / 1e3:.1f} us")
for n, a in agg.items(): print(f"  {n:42s} x{a[0]:4d}  {a[1] / 1e3:8.1f} us  gaps before {a[2] / 1e3:8.1f} us")
