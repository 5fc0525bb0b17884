#!/bin/bash
# tools/valu_calibrate.sh [tag]  -- ON THE GPU BOX: what a VALU wave-instruction of each kind costs a SIMD on THIS chip, and what the busy
# counters make of it.  tools/ubench/valu_rate (independent chains of ONE instruction kind, 8 waves per SIMD) is timed, then run again
# under rocprofv3 with SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE: per kind, cycles per wave-instruction
# per SIMD from the duration x the measured clock, and SQ_ACTIVE_INST_VALU ticks per instruction.  -> gpurun_out/<tag>_valu_calibration.json
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/valu_cal
hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_rate tools/ubench/valu_rate.hip || exit 1
./tools/ubench/valu_rate > gpurun_out/valu_cal/plain.txt
cat gpurun_out/valu_cal/plain.txt
for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d gpurun_out/valu_cal/pmc_$N -o pmc -- ./tools/ubench/valu_rate > /dev/null 2>&1
done
python - <<PY
import csv, glob, json, collections, re
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/valu_cal/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_valu<(\d+)>", r["Kernel_Name"])
        if not m: continue
        rows[(int(m.group(1)), int(r["Dispatch_Id"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[(int(m.group(1)), int(r["Dispatch_Id"]))]["_ns"] = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"])]
names = {0: "v_fma_f32", 1: "v_mul_f32", 2: "v_add_f32", 3: "v_lshlrev_b32", 4: "v_bfe_i32", 5: "v_cvt_f32_i32", 6: "v_cndmask_b32", 7: "v_cmp_lt_f32 vcc",
         10: "v_cmp_lt_f32 sgpr", 8: "v_pk_mul_f32", 9: "v_pk_add_f32", 11: "v_fma_f64", 12: "v_mul_f64", 13: "v_add_f64", 14: "v_add_u32"}
by_op = collections.defaultdict(dict)
for (op, disp), c in rows.items():
    # the long launch of each kind (iters 200) is the second dispatch of the kind in a pass: keep the one with more instructions
    tot = {k: sum(v) for k, v in c.items()}
    for k, v in tot.items():
        by_op[op].setdefault(k, []).append(v)
out = {}
for op, c in sorted(by_op.items()):
    big = {k: max(v) for k, v in c.items()}   # the 200-iteration launch
    insts = big.get("SQ_INSTS_VALU")
    e = {"counters": big}
    if insts and big.get("GRBM_GUI_ACTIVE"):
        cycles = big["GRBM_GUI_ACTIVE"] / 8.0
        simds = 256 * 4
        e["cycles_per_wave_instruction_per_simd"] = cycles * simds / insts
        e["sq_active_inst_valu_per_instruction"] = big.get("SQ_ACTIVE_INST_VALU", 0) / insts
        e["valu_busy_by_4x_counter"] = 4.0 * big.get("SQ_ACTIVE_INST_VALU", 0) / simds / cycles
    out[names.get(op, str(op))] = e
json.dump({"note": "tools/valu_calibrate.sh: one instruction kind per launch, 8 waves per SIMD, independent chains; cycles = GRBM_GUI_ACTIVE / 8 XCDs of the same launch",
           "plain_timing": open("gpurun_out/valu_cal/plain.txt").read().splitlines(), "kinds": out}, open("gpurun_out/${TAG}_valu_calibration.json", "w"), indent=1)
for k, e in out.items():
    print("%-20s %6.2f cycles/instr/SIMD   ACTIVE_INST_VALU per instr %.3f   '4 x counter' busy %.2f" % (k, e.get("cycles_per_wave_instruction_per_simd", 0), e.get("sq_active_inst_valu_per_instruction", 0), e.get("valu_busy_by_4x_counter", 0)))
PY
rm -rf gpurun_out/valu_cal
