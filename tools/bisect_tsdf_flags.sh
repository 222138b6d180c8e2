for rf in 0 1; do for fs in 0 1; do for fc in 0 1; do
  echo "== ROW_FAR=$rf FRAME_SKIP=$fs FAST_COLOUR=$fc"
  HIVE_TSDF_ROW_FAR=$rf HIVE_TSDF_FRAME_SKIP=$fs HIVE_TSDF_FAST_COLOUR=$fc timeout -k 10 120 python -m pytest tests/test_pipeline_gpu.py -x -q -m gpu -k partition_property 2>&1 | tail -1
done; done; done
