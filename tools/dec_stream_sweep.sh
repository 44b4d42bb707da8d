S="B16@32k,B8@32k,tp8 B64,B4@32k,B2@32k,G32 B16"
for i in 1 2 3; do
echo "== stream"; python tools/kbench.py decode --only "$S" 2>&1 | grep "splits="
echo "== legacy"; python tools/kbench.py decode --variant 524288 --only "$S" 2>&1 | grep "splits="
done
