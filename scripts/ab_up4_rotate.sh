# same-box A/B of the fused-phase kernel's K rotation (conv_up4.hip): scripts/time_up4.py with CAGC_UP4_ROTATE = number of phases
for st in ${PHASES:-0 2 4 8 0 4}; do echo ROTATE $st; CAGC_UP4_ROTATE=$st timeout 200 python scripts/time_up4.py 2>&1 | grep -v "amdgpu.ids\|@16^2\|dgrad 512<-512 @32"; done
