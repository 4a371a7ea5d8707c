for rep in 1 2; do
for lib in tools/_ab/libvtmc_diag_head.so tools/_ab/libvtmc_diag.so; do
  for ab in 0 1; do
    echo "== $lib ablate=$ab signs=1: $(VTMC_FILL_SIGNS=1 VTMC_LIB=$lib python tools/fill_only.py fbm8 $ab)"
  done
done
done
