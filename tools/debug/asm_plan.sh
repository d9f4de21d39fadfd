#!/bin/bash
# the planner's translation unit as gfx950 assembly, the chain functions cut out: /tmp/p4asm/{cw,kw}.s
mkdir -p /tmp/p4asm && cd /tmp/p4asm || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -I/root/repo/include -o plan.s /root/repo/erlangnetwork-gnsslib-sdr_amd/csrc/gnsscorr_plan.hip 2>&1 | grep -E "error" -A8 | head -30
awk '/^_ZN12_GLOBAL__N_115plan4_code_waveILi11ELi8EEEvddiiPKNS_8Plan4JobEi:/{p=1} p{print} /\.Lfunc_end.*plan4_code_waveILi11ELi8/{p=0}' plan.s > cw.s
awk '/^_ZN12_GLOBAL__N_114plan4_car_wave/{p=1} p{print} /\.Lfunc_end.*plan4_car_wave/{p=0}' plan.s > kw.s
wc -l cw.s kw.s
