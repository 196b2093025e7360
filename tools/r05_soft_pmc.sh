cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in tools/libhead.so mvtn_amd/libmvr_hip.so; do
  tag=$(basename $lib .so)
  MVR_LIB=$lib timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/softpmc_$tag -- python3 bench.py --workload soft_fwdbwd --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - gpurun_out/softpmc_$tag $tag <<'PY'
import csv,glob,sys
acc={}
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mesh_soft_forward" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v)/len(v)/1e6,3) for k,v in acc.items()})
PY
done
