# the whole GPU suite twice on one box (closing passes: profiles/r06/README.md); usage: suite_twice.sh <tag>
t=${1:-x}
for i in 1 2; do
  python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/suite_${t}_$i.txt 2>&1
  echo "run $i rc=$? $(grep -E 'passed|failed' gpurun_out/suite_${t}_$i.txt | tail -1)"
done
