# kNN scan as one GEMM (Q = 256 x N = 100 000 x D = 6144): tile order (row panels of one bank tile first vs the projections' order) x ring tile
# Usage (GPU box): bash scripts/knn_gemm_ab.sh > gpurun_out/r05_knn_gemm_ab.log
for nf in 0 1; do for ring in -1 1 2 3; do
  echo "ASTTS_KNN_GEMM_N_FIRST=$nf ASTTS_GEMM_RING=$ring"
  ASTTS_KNN_GEMM_N_FIRST=$nf ASTTS_GEMM_RING=$ring KNN_N=100000 KNN_D=6144 KNN_Q=256 KNN_ITERS=20 python scripts/knn_small.py 2>&1 | grep -v "experiment switch" | tail -1
done; done
echo "D=768:"
for nf in 0 1; do ASTTS_KNN_GEMM_N_FIRST=$nf KNN_N=100000 KNN_D=768 KNN_Q=256 KNN_ITERS=20 python scripts/knn_small.py 2>&1 | tail -1; done
