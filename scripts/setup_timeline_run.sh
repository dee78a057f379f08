R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/r05_setup_tl -o run --output-format csv -- python3 $R/scripts/setup_prof.py coil 128 > $R/gpurun_out/r05_setup_tl.log 2>&1
cd $R
python3 scripts/setup_timeline.py gpurun_out/r05_setup_tl 15 > gpurun_out/r05_setup_timeline_coil128.txt 2>&1
cat gpurun_out/r05_setup_timeline_coil128.txt
rm -rf gpurun_out/r05_setup_tl
