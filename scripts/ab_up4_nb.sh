# same-box A/B of the fused-phase kernel's workgroup shape (conv_up4.hip): CAGC_UP4_NB = 2 (two workgroups per CU) / 4 (one)
for nb in ${NBS:-2 4 2 4}; do echo NB $nb ROTATE ${CAGC_UP4_ROTATE:-0}; CAGC_UP4_NB=$nb timeout 200 python scripts/time_up4.py 2>&1 | grep -v "amdgpu.ids\|@16^2\|dgrad 512<-512 @32"; done
