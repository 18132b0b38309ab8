set -x
O=gpurun_out/r03a; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_inplace.log 2>&1; echo "rc=$?" >> $O/smoke_inplace.log
rm -rf /tmp/treecopy && mkdir -p /tmp/treecopy && cp -rL . /tmp/treecopy/repo2 2>/dev/null
(cd /tmp/treecopy/repo2 && pwd && python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke_copied_tree.log 2>&1; echo "rc=$?" >> $O/smoke_copied_tree.log
python tools/strong_proxy.py > $O/strong_proxy.jsonl 2> $O/strong_proxy.err
python bench.py --no-cpu --no-configs --steps 40 --warmup 5 > $O/bench_short.json 2> $O/bench_short.err
tail -3 $O/*.log; cat $O/strong_proxy.jsonl | cut -c1-300
