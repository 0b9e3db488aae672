"""Summarise a rocprofv3 --pmc run: per kernel name, launches and mean counter value; kernel FAMILIES (template variants of
one kernel: lm_gemv<...>, gemm_ring<...>) are also aggregated.  Optional 2nd argument: write {family: hbm_bytes_per_launch}
JSON from FETCH_SIZE (KB x 1024 x 2: gfx950 reports half of a 16-B-per-lane coalesced stream, MI355X_MICROARCH.md, HBM)."""
import csv, glob, sys, collections, json, re
d = sys.argv[1]
def demangle(n):
    """kernels with explicit parameters show up mangled in rocprofv3's CSVs (and binutils' c++filt does not know _Float16)"""
    m = re.match(r"_ZN5astts(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1))
    name, rest, args = n[m.end():m.end() + ln], n[m.end() + ln:], []
    if rest.startswith("I"):
        rest = rest[1:]
        while True:
            a = re.match(r"L([ib])(n?\d+)E", rest)
            if not a:
                break
            args.append(a.group(2))
            rest = rest[a.end():]
    return name + ("<" + ", ".join(args) + ">" if args else "")
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
fam = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in files:
    for r in csv.DictReader(open(f)):
        k = demangle(r['Kernel_Name']).split('(')[0].replace('void ', '').replace('astts::', '')[-56:]
        a = agg[k][r['Counter_Name']]
        a[0] += 1; a[1] += float(r['Counter_Value'])
        m = re.search(r'(lm_gemv|lm_attn|gemm_ring|gemm_tile|gemm_skinny16|attn_mha_flash|tfm_attn_fused|tfm_ffn_fused|knn_scan)', r['Kernel_Name'])
        if m:
            b = fam[m.group(1)][r['Counter_Name']]
            b[0] += 1; b[1] += float(r['Counter_Value'])
for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:30]:
    print(k, {c: (n, round(s / n, 2)) for c, (n, s) in cs.items()})
print('-- families (all template variants)')
for k, cs in sorted(fam.items()):
    print(k, {c: (n, round(s / n, 2)) for c, (n, s) in cs.items()})
if len(sys.argv) > 2:
    out = {k: {"launches_sampled": cs['FETCH_SIZE'][0], "hbm_bytes_per_launch": int(cs['FETCH_SIZE'][1] / cs['FETCH_SIZE'][0] * 1024 * 2)}
           for k, cs in fam.items() if 'FETCH_SIZE' in cs}
    json.dump(out, open(sys.argv[2], 'w'), indent=1)
