export TMPDIR=/tmp
for cfg in "atrium260k --spp 16" "blob7m --spp 4" "terrain1m --spp 1"; do
name=$(echo $cfg | cut -d' ' -f1)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_$name -o k -- python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu --no-count --no-others > gpurun_out/kt_$name.json 2> gpurun_out/kt_$name.log || exit 1
done
