#!/bin/bash
# md5 of the embedded gfx950 code of every object under latent-diffusion_amd/build: run before and after an edit that is meant
# to leave the device code untouched (comments, host-side checks, LDM_LOCKSTEP markers) and compare.
for f in "$(dirname "$0")"/../latent-diffusion_amd/build/*.o; do
  /opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=/tmp/_fat.bin "$f" /tmp/_fat_copy.o 2>/dev/null && echo "$(md5sum < /tmp/_fat.bin | cut -d' ' -f1) $(basename "$f")"
done
