cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/pmc_linear_shapes.py time > gpurun_out/lin_times.json 2> /tmp/t.err || tail -3 /tmp/t.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/lf -- python tools/pmc_linear_shapes.py run > /tmp/lf.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/lw -- python tools/pmc_linear_shapes.py run > /tmp/lw.log 2>&1
python tools/pmc_linear_shapes.py fold "$(ls -S /tmp/lf/*/*counter_collection.csv | head -1)" "$(ls -S /tmp/lw/*/*counter_collection.csv | head -1)" gpurun_out/lin_times.json > gpurun_out/lin_pmc.txt 2>&1
cat gpurun_out/lin_pmc.txt
