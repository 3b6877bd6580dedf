mkdir -p gpurun_out/xcd
for W in "courtyard_like_10M_1920x1080_121spp --samples-sqrt 6" "sponza_like_1920x1080_256spp_envmap_is --samples-sqrt 8"; do
  for V in 16 ; do
    N=$(echo $W | cut -d_ -f1)
    timeout -k 10 300 python bench.py --workload $W --variant $V --steps 2 --warmup 1 --no-secondary --cpu-seconds 2 > gpurun_out/xcd/${N}_v${V}_plain.json 2>/dev/null || exit 1
    WPT_XCD_BANDS=1 timeout -k 10 300 python bench.py --workload $W --variant $V --steps 2 --warmup 1 --no-secondary --cpu-seconds 2 > gpurun_out/xcd/${N}_v${V}_bands.json 2>/dev/null || exit 1
  done
done
echo done
