#!/bin/bash
# Calibration (GPU box): what do the per-class VALU counters report for instructions whose count is known?
# tools/microbench/valu_cycles launches, per instruction, kernels of 256 x w workgroups x 4 waves x 8192 x 16 instructions.
export TMPDIR=/tmp
rm -rf gpurun_out/pmc_cal
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d gpurun_out/pmc_cal -- tools/microbench/valu_cycles > gpurun_out/pmc_cal.out 2>&1 || exit 1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_cal/*/*counter_collection.csv')[0]
rows=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    rows.setdefault(r['Dispatch_Id'],{})[r['Counter_Name']]=float(r['Counter_Value'])
    rows[r['Dispatch_Id']]['grid']=int(r['Grid_Size']); rows[r['Dispatch_Id']]['name']=r['Kernel_Name']
names=["v_fma_f32","v_pk_fma_f32","v_mul_lo_u32","v_xor_b32","v_cvt_f32_u32","v_max_f32","v_add_f32","v_mul_f32","v_lshrrev_b32","v_alignbit_b32","v_and_or_b32","v_rsq_f32","v_fmac_f32","v_mad_u32_u24","v_pk_mul_f32","v_mov_b32","v_cvt_f32_ubyte1","v_cvt_f32_ubyte3","v_xor_b32_sdwa","v_pk_fma_f32 op_sel","v_pk_add_f32 op_sel","v_fma_f32 |x| -y","v_max3_f32","v_bcnt_u32_b32"]
ds=list(rows.values())
# 8 dispatches per instruction (warm-up + measured, w = 1, 2, 4, 8): take the last (w = 8, measured)
for k,n in enumerate(names):
    d=ds[k*8+7]; insts=d['grid']/64*8192*16
    print(f"{n:22s}", {c[14:]:round(v/insts,3) for c,v in d.items() if c.startswith('SQ_INSTS_VALU_')}, 'VALU/inst', round(d['SQ_INSTS_VALU']/insts,3))
PY
