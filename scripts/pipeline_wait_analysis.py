"""Where do the decode chains wait inside the pipelined benchmark?  Input: a rocprofv3 --kernel-trace CSV of bench.py.
For every decode-step kernel (lm_gemv / lm_attn / ras_sample) on its queue: wait = start - end of the previous kernel of the
same queue (a dependent launch chain: ~1.5 us when nothing is in the way).  The wait is attributed to the render-stream kernel
that covers most of [previous end, start).  Output: totals per blocking kernel + the decode kernels' own durations by what ran beside.

  rocprofv3 --kernel-trace -d /tmp/t -o t --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-24khz --no-cobatch
  python3 scripts/pipeline_wait_analysis.py /tmp/t"""
import bisect
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(lm_gemv|lm_attn|ras_sample|tfm_attn_fused|tfm_ffn_fused|rconv_lds|attn_mha_flash|gemm_ring|gemm_skinny\w*|conv_lds\w*|snake\w*|\w+)", name.split("(")[0].split("<")[0].replace("astts::", ""))
    n = name
    for pat in ("lm_gemv", "lm_attn", "ras_sample", "tfm_attn_fused", "tfm_ffn_fused", "rconv_lds", "attn_mha_flash", "gemm_ring", "gemm_skinny", "conv_lds", "layernorm", "attn_relpos"):
        if pat in n:
            return pat
    return (m.group(1) if m else n)[:28]


def main(root):
    path = [p for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)][0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], short(r["Kernel_Name"])))
    rows.sort()
    t_end = rows[-1][1]
    decode = {"lm_gemv", "lm_attn", "ras_sample"}
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    qkind = {}
    for q, rs in byq.items():
        nd = sum(1 for r in rs if r[3] in decode)
        qkind[q] = "decode" if nd > 0.9 * len(rs) else ("render" if any(r[3].startswith("tfm_") for r in rs) else "other")
    print("queues:", {q: (qkind[q], len(rs)) for q, rs in byq.items()})
    render = sorted(r for q, rs in byq.items() if qkind[q] == "render" for r in rs)
    rstart = [r[0] for r in render]
    # only the steady state: the last 60 % of the trace (warm-up, calibration and checks come first)
    t0 = rows[0][0] + int(0.4 * (t_end - rows[0][0]))

    def covering(a, b):
        """render kernel with the largest overlap with [a, b)"""
        i = max(bisect.bisect_right(rstart, a) - 1, 0)
        best, bo = "(idle)", 0
        while i < len(render) and render[i][0] < b:
            o = min(b, render[i][1]) - max(a, render[i][0])
            if o > bo:
                best, bo = render[i][3], o
            i += 1
        return best if bo > 0.3 * (b - a) else "(idle)"

    wait_by = collections.defaultdict(lambda: [0, 0.0])
    dur_by = collections.defaultdict(lambda: [0, 0.0])
    tot_wait = tot_dur = 0.0
    n = 0
    for q, rs in byq.items():
        if qkind[q] != "decode":
            continue
        for p, k in zip(rs, rs[1:]):
            if k[0] < t0 or k[0] - p[1] > 200_000:      # a new decode call: not a chain step
                continue
            w = (k[0] - p[1]) / 1000.0
            blk = covering(p[1], k[0])
            wait_by[blk][0] += 1
            wait_by[blk][1] += w
            beside = covering(k[0], k[1])
            dur_by[(k[3], beside)][0] += 1
            dur_by[(k[3], beside)][1] += (k[1] - k[0]) / 1000.0
            tot_wait += w
            tot_dur += (k[1] - k[0]) / 1000.0
            n += 1
    print(f"decode launches analysed: {n}; mean wait before a launch {tot_wait / n:.2f} us, mean duration {tot_dur / n:.2f} us")
    print("wait before a decode launch, by the render kernel covering the wait:")
    for blk, (c, w) in sorted(wait_by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {blk:18s} launches {c:7d} ({100.0 * c / n:5.1f} %)  mean wait {w / c:6.2f} us  share of all waiting {100.0 * w / tot_wait:5.1f} %")
    print("decode kernel durations by what ran beside them:")
    for (k, b), (c, d) in sorted(dur_by.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"  {k:11s} beside {b:18s} launches {c:7d}  mean {d / c:6.2f} us")
    # render stream occupancy in the window
    busy = sum(min(r[1], t_end) - max(r[0], t0) for r in render if r[1] > t0)
    print(f"render stream busy {100.0 * busy / (t_end - t0):.1f} % of the analysed window ({(t_end - t0) / 1e6:.1f} ms)")
    rk = collections.defaultdict(lambda: [0, 0.0])
    for r in render:
        if r[0] >= t0:
            rk[r[3]][0] += 1
            rk[r[3]][1] += (r[1] - r[0]) / 1000.0
    for k, (c, d) in sorted(rk.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  render {k:18s} launches {c:6d} mean {d / c:6.2f} us total {d / 1000.0:7.2f} ms")


if __name__ == "__main__":
    main(sys.argv[1])
