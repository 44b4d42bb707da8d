S="B1@128k,B1@32k,B1@8k,B2@32k,B4@32k"
for i in 1 2; do
echo "== product (grid heuristics for B=1)"; python tools/kbench.py decode --rotate --only "$S" 2>&1 | grep "splits="
echo "== lab: two K/V register sets per wave (PF = 2), grid heuristics"; python tools/kbench.py decode --variant $((262144 + 524288)) --rotate --only "$S" 2>&1 | grep "splits="
done
for n in 32 48 64 96; do echo "== PF=2, $n splits"; python tools/kbench.py decode --variant $((262144 + 524288)) --rotate --only "B1@128k,B1@32k" --splits $n 2>&1 | grep "splits="; done
