#!/bin/bash
# Where do a kernel's wave-cycles go?  One rocprofv3 --pmc pass (8 SQ slots) over tools/gma_stage_bench.py (or "$@"), summarised per kernel:
# WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES (quad-cycles).
# usage: tools/pmc_wave_states.sh [command ...]
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD=${@:-python tools/gma_stage_bench.py}
rm -rf gpurun_out/pmc_ws
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES --kernel-trace -d gpurun_out/pmc_ws -o pmc -- $CMD > /dev/null 2> gpurun_out/pmc_ws.err
python - <<'PY'
import sqlite3, glob, collections, sys
sys.path.insert(0, "tools")
from rocpd_summary import demangle
db = glob.glob("gpurun_out/pmc_ws/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"""select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pmc} e join {info} i on e.pmc_id = i.id
        join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id group by s.kernel_name, i.name"""
acc = collections.defaultdict(dict); nd = {}
for k, c, v, n in con.execute(q):
    acc[k][c] = v; nd[k] = n
rows = sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]
print("| kernel | dispatches | wave quad-cycles | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | VALU issuing | LDS issuing | LDS issue stall | waves resident per CU (WAVE / BUSY_CU) |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, c in rows:
    w = c.get("SQ_WAVE_CYCLES", 0) or 1
    f = lambda n: f"{c.get(n, 0) / w:.2f}"
    busy = c.get("SQ_BUSY_CU_CYCLES", 0) or 1
    print(f"| `{str(demangle(k))[:70]}` | {nd[k]} | {w:.3g} | {f('SQ_WAIT_ANY')} | {f('SQ_WAIT_INST_ANY')} | {f('SQ_ACTIVE_INST_ANY')} | {f('SQ_ACTIVE_INST_VALU')} | {f('SQ_ACTIVE_INST_LDS')} | {f('SQ_WAIT_INST_LDS')} | {4 * w / busy:.1f} |")
PY
