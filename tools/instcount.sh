R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for v in . NOFRAG SKIP_TEX; do
  L=$R/mesh2splat_amd/_build/$v/libm2s_hip.so
  M2S_LIB_PATH=$L timeout 100 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $R/gpurun_out/ic_$v -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sync-steps > /dev/null 2>&1
  echo "== $v"; python $R/tools/pmc_summary.py $R/gpurun_out/ic_$v/f_counter_collection.csv | python -c "
import json,sys; d=json.load(sys.stdin)['m2s::k_fused']; w=d['SQ_WAVES']; print({k:round(v/w,1) for k,v in d.items()})"
done
