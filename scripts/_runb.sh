cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_knn_gpu.py -m gpu -q 2>&1 | tail -3
rm -rf /tmp/pk; KNN_ITERS=300 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pk -o k --output-format csv -- python3 scripts/knn_small.py > /dev/null 2>&1
python3 -c "
import csv,sys,glob
for r in csv.DictReader(open(glob.glob('/tmp/pk/**/*kernel_stats.csv',recursive=True)[0])):
    if 'knn_' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'])
"
timeout 100 python scripts/knn_small.py
KNN_N=100000 KNN_Q=8 KNN_ITERS=200 timeout 100 python scripts/knn_small.py
KNN_N=100000 KNN_Q=256 KNN_ITERS=100 timeout 100 python scripts/knn_small.py
