set -e
cd $GRAFT_REPO_ROOT
g++ -std=c++17 -O2 -Iinclude examples/bench_yolo.cpp -Lsimpleinfer_amd -lsimpleinfer_amd -lsi_hip -Wl,-rpath,$PWD/simpleinfer_amd -o /tmp/bench_yolo
python - <<'PY'
import sys
sys.path.insert(0, '.')
from simpleinfer_amd import modelgen as mg
mg.build_yolov5s(2, 160).save('/tmp/s.param', '/tmp/s.bin')
mg.build_yolov5s(8, 640).save('/tmp/b8.param', '/tmp/b8.bin')
PY
/tmp/bench_yolo /tmp/s.param /tmp/s.bin 50
/tmp/bench_yolo /tmp/b8.param /tmp/b8.bin 20
