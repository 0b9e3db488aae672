"""Copy a rocprofv3 kernel_stats CSV with the kernel names readable: kernels with explicit parameters are listed mangled, and
binutils' c++filt does not know _Float16 (DF16_), so the astts:: names are unmangled here as far as the template arguments."""
import csv, re, sys


def pretty(n):
    m = re.match(r"_ZN5astts(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1))
    name = n[m.end():m.end() + ln]
    rest = n[m.end() + ln:]
    args = []
    if rest.startswith("I"):
        rest = rest[1:]
        while True:
            a = re.match(r"L([ib])(n?\d+)E", rest)
            if not a:
                break
            v = a.group(2).replace("n", "-")
            args.append(("true" if v == "1" else "false") if a.group(1) == "b" else v)
            rest = rest[a.end():]
    return "astts::" + name + ("<" + ", ".join(args) + ">" if args else "") + "  [" + n + "]"


rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(open(sys.argv[2], "w", newline=""), quoting=csv.QUOTE_NONNUMERIC)
for i, r in enumerate(rows):
    if i and r:
        r[0] = pretty(r[0])
        r[1:] = [float(x) if "." in x else int(x) for x in r[1:]]
    w.writerow(r)
