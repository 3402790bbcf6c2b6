#!/bin/bash
# usage: tools/isa.sh <name> "<extra -D flags>"  -> build/isa/<name>.s (device assembly of the fp32 kernel TU, gfx950) + a per-kernel summary
# (instruction counts by class, VGPRs, scratch, LDS) of the kernels whose mangled name matches $3 (default: the traversal and camera kernels)
set -e
cd "$(dirname "$0")/../rs_ray_toy_amd/csrc"
out=../../build/isa; mkdir -p $out
[ -n "$ISA_KEEP" ] && [ -f $out/$1.s ] || /opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -Wno-unused-function -Wno-unused-command-line-argument -I../../include -Ihost -Idevice -fno-hip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize $2 --cuda-device-only -S device/rrt_f32.hip -o $out/$1.s
python3 - "$out/$1.s" "${3:-k_trace_pt_f32|k_trace_tiles_f32|k_trace_pairs_f32|k_shadow_lists_f32|k_raygen_main_f32|k_shade_path}" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*\.end_amdhsa_kernel", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not pat.search(name): continue
    ins = re.findall(r"^\s+([sv]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)", body, re.M)
    def c(p): return sum(1 for i in ins if re.match(p, i))
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body); sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body); lds = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", body)
    print(f"{name[:70]:70s} insts {len(ins):5d} valu {c('v_'):5d} (pk {c('v_pk_'):3d} fma {c('v_fma|v_fmac'):3d} minmax {c('v_min|v_max|v_med'):3d} cmp {c('v_cmp'):3d} cnd {c('v_cndmask'):3d}) salu {c('s_'):5d} ds {c('ds_'):3d} vmem {c('global_|buffer_|flat_'):3d} scratch {c('scratch_'):3d} vgpr {vg.group(1) if vg else '?'} priv {sc.group(1) if sc else '?'} lds {lds.group(1) if lds else '?'}")
PY
