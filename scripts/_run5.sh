D=/tmp/e2e; rm -rf $D; mkdir -p $D
python3 scripts/e2e_setup.py $D 1024 > /dev/null 2>&1
run() { A=$1; tag=$2; shift 2; env "$@" python3 -m dipoorlet_amd -M $D/r50.onnx -I $D/calib -N 1024 -A $A -D trt -O $D/out_$tag --calib_batch 32 --skip_profiling --timing_json $D/t_$tag.json > /tmp/cli_$tag.log 2>&1 || tail -5 /tmp/cli_$tag.log; echo "$tag: $(cat $D/t_$tag.json)"; }
run hist hist X=1
run mse mse_tail X=1
run hist hist2 X=1
run mse mse2 X=1
