"""Per-geometry table of the GEMM launches of a bench run made with LHG_PROFILE_LOG=<csv>.
usage: python tools/layer_table.py <csv> [products per multiply-add, default 6]
columns of the log: kernel(0 gather / 1 wgrad), M, rows(N pad | m pad), cols(Ci | n pad), Ci, Co, taps, istep, ostep, Hi, variant, flops, ms"""
import collections, csv, sys
rows = list(csv.reader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    key = tuple(r[:11])
    a = agg.setdefault(key, [0, 0.0, float(r[11])])
    a[0] += 1
    a[1] += float(r[12])
tot = {"0": 0.0, "1": 0.0}
for k, (n, ms, fl) in agg.items():
    tot[k[0]] += ms
print("kern      M  rows  cols    Ci    Co taps is os   Hi  var    n   avg us  TFLOP/s  share")
for k, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%4s %7s %5s %5s %5s %5s %4s %2s %2s %4s %4s %4d %8.1f %8.1f %5.1f%%" % (*k, n, ms / n * 1e3, fl * n / ms / 1e9, 100 * ms / tot[k[0]]))
