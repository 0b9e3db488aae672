"""Summarise a rocprofv3 --pmc run: per kernel name, launches and mean counter value; kernel FAMILIES (template variants of
one kernel: lm_gemv<...>, gemm_ring<...>) are also aggregated.  Optional 2nd argument: write {family: hbm_bytes_per_launch}
JSON from FETCH_SIZE (KB x 1024 x 2: gfx950 reports half of a 16-B-per-lane coalesced stream, MI355X_MICROARCH.md, HBM)."""
import csv, glob, sys, collections, json, re
d = sys.argv[1]
files = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
fam = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][-48:]
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
