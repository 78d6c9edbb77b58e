# same-box timing of conv_up25.hip build variants (scripts/build_up25_variants.sh): LIBS="libcagc_hip.so libcagc_hip_nost.so ..."
for lib in ${LIBS:-libcagc_hip.so}; do echo LIB $lib; LIB=$lib ONLY=up25 timeout 200 python scripts/time_up4.py 2>&1 | grep -v "amdgpu.ids\|@16^2\|dgrad 512<-512 @32"; done
