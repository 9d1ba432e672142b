cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/k2
P="--no-cpu-baseline --profile-steps 0 --model_type VGG_PONG_LAYERNECK --model_in_channels 1 --model_z_channels 16 --model_keypoints 4 --size 84 --batch 256"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k2 -o k -- python /root/repo/bench.py $P --steps 8 --warmup 2 --streams 0 > /dev/null 2>&1
cp $(find /tmp/k2 -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/r02k_cfg2_kernel_stats.csv
python -m pytest tests/test_gpu_conv_mfma.py -q -k small 2>&1 | tail -1
