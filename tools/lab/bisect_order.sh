T=tests/test_gpu_mobilenet.py::test_mobilenet_step_matches_oracle
for m in test_gpu_conv_ops test_gpu_winograd test_gpu_split_engine test_gpu_comm test_gpu_inception test_gpu_detection test_gpu_postprocess; do
  echo "=== $m"; python -m pytest tests/$m.py $T -x -q -m gpu 2>&1 | tail -3
done
echo "=== mobilenet whole module"; python -m pytest tests/test_gpu_mobilenet.py -x -q -m gpu 2>&1 | tail -3
