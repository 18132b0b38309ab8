O=gpurun_out/r03p; mkdir -p $O
python tools/ab_small.py ab_libs/base.so ab_libs/fake.so --sizes=1048576,262144 > $O/fake.jsonl 2>$O/ab.err
cat $O/fake.jsonl
