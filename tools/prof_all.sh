set -uo pipefail
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/f1_tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/f1_tests.txt
for spec in "glass f64" "glass f32" "random f64" "random_glass f64" "default f64"; do
  set -- $spec
  timeout -k 10 900 ./tools_prof.sh r03_$1_$2 --scene $1 --precision $2 > gpurun_out/prof_r03_$1_$2.txt 2>&1 || { echo "prof $1 $2 failed"; tail -5 gpurun_out/prof_r03_$1_$2.txt; exit 1; }
  grep "== derived" gpurun_out/prof_r03_$1_$2.txt | cut -c1-220
done
