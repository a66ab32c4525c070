# WRITE_SIZE / FETCH_SIZE of the per-track kernel (separate passes), previous library (msckf_mono_amd/lib_ab/libmsckf_hip_prev.so) and current one
cd /tmp && export TMPDIR=/tmp
CM="--steps 6 --warmup 2 --no-cpu-baseline --no-early-accept-pass --no-upload-pass --no-other-configs --repeats 1 --streams 1"
for v in prev cur; do
  if [ $v = prev ]; then export MSCKF_HIP_LIB=/root/repo/msckf_mono_amd/lib_ab/libmsckf_hip_prev.so; else unset MSCKF_HIP_LIB; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    timeout 240 rocprofv3 --pmc $c --kernel-trace -d /tmp/fw$v$c -o r -- python /root/repo/bench.py $CM > /tmp/fw$v$c.log 2>&1
  done
  timeout 100 python /root/repo/scripts/rocpd_pmc.py /tmp/fw_$v.md $(find /tmp/fw${v}WRITE_SIZE /tmp/fw${v}FETCH_SIZE -name "*.db") > /dev/null 2>&1
  grep -i "k_feature" /tmp/fw_$v.md | cut -c1-200 | sed "s/^/$v /"
done
