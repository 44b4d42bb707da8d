S="B16@32k,B8@32k,B4@32k,B2@32k,tp8 B64,B64@8k,B256@2k,B8@128k"
for i in 1 2; do
echo "== lab, fair-share priority ON"; python tools/kbench.py decode --variant 2097152 --rotate --only "$S" 2>&1 | grep "splits="
echo "== lab, fair-share priority OFF"; python tools/kbench.py decode --variant 8388608 --rotate --only "$S" 2>&1 | grep "splits="
done
