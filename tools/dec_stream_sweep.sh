S="B1@128k,B1@32k,B2@32k,B4@32k,B16@32k"
echo "== legacy, one cache (Infinity Cache warm)"; python tools/kbench.py decode --variant 524288 --only "$S" 2>&1 | grep "splits="
echo "== legacy, rotating caches"; python tools/kbench.py decode --variant 524288 --only "$S" --rotate 2>&1 | grep "splits="
echo "== product default, rotating"; python tools/kbench.py decode --only "$S" --rotate 2>&1 | grep "splits="
for n in 24 48 64 96 128 192; do echo "== legacy rotating, $n splits"; python tools/kbench.py decode --variant 524288 --only "B1@128k,B1@32k" --rotate --splits $n 2>&1 | grep "splits="; done
for n in 48 96 192 384; do echo "== stream rotating, $n workgroups per head"; python tools/kbench.py decode --only "B1@128k,B1@32k" --rotate --splits -$n 2>&1 | grep "splits="; done
