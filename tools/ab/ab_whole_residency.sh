# residency of a whole-batch launch of the one-wave step kernels (LDS padding to whole rounds): CC4_WHOLE_RESIDENCY = 0 (off) / unset (auto) / n per CU
for r in 0 auto 18 16 14 12; do
  if [ $r = auto ]; then unset CC4_WHOLE_RESIDENCY; else export CC4_WHOLE_RESIDENCY=$r; fi
  for n in 8192 6144 12288; do echo "residency=$r"; python tools/host_step_probe.py $n 60 2>&1 | grep "device policy"; done
done
