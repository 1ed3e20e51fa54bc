cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 0 auto; do
  export RTS_XCD_AFFINE=$m
  for p in tcc fetch write; do
    case $p in tcc) C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum";; fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
    OUT=$ROOT/gpurun_out/pmc_r04i_c4_affine_$m/$p; mkdir -p $ROOT/gpurun_out/pmc_r04i_c4_affine_$m
    timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/trace_bench.py c4 6 > $OUT.log 2>&1
    find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_agent_info.csv" -delete
    for f in $(find $OUT -name "*_counter_collection.csv"); do { head -1 $f; grep k_trace $f; } > $f.tmp && mv $f.tmp $f; done
    tail -1 $OUT.log
  done
done
python3 - <<'PY'
import csv, glob, os, collections
root=os.environ["GRAFT_REPO_ROOT"]
for m in ("0","auto"):
    for p in ("tcc","fetch","write"):
        fs=glob.glob(os.path.join(root,"gpurun_out","pmc_r04i_c4_affine_%s"%m,p,"**","*_counter_collection.csv"),recursive=True)
        if not fs: print(m,p,"no file"); continue
        d=collections.OrderedDict()
        for r in csv.DictReader(open(fs[0])):
            k=(int(r["Dispatch_Id"]), "coop" if r["Kernel_Name"].split("<",1)[-1].split(",")[3].strip()=="true" else "ord")
            d.setdefault(k,{})[r["Counter_Name"]]=d.setdefault(k,{}).get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
            d[k]["_ms"]=(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e6
        rows=list(d.items())[2:]   # skip the first launches
        agg=collections.defaultdict(list)
        for (disp,kind),v in rows:
            for c,x in v.items(): agg[(kind,c)].append(x)
        print("affine",m,p,{k: round(sum(v)/len(v),3) for k,v in agg.items()})
PY
