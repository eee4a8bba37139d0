mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_fuzz_parity.py -m gpu -q -k "ill_conditioned" > gpurun_out/r6/t8_tests.log 2>&1; echo "rc $?" >> gpurun_out/r6/t8_tests.log; tail -3 gpurun_out/r6/t8_tests.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6/prof_exact_c4b -o exact_c4 -- python3 $R/tools/exact_stack_prof.py c4 48 576 256 16 50 > $R/gpurun_out/r6/t8_prof_c4.log 2>&1
cd $R
python3 - <<'PY'
import sqlite3,glob
name=glob.glob('gpurun_out/r6/prof_exact_c4b/*.db')[0]
db=sqlite3.connect(name)
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
for r in db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc limit 8"):
    print(f"  {r[0][:80]:80s} n={r[1]:6d} total_us={r[2]:10.1f} avg_us={r[3]:8.2f}")
PY
grep "us/iter" gpurun_out/r6/t8_prof_c4.log
