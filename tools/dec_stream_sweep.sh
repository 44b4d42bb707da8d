python tools/ragged_decode_probe.py 2>&1 | grep "==\|DEVICE\|512 work\|768 work\|256 work\|EQUAL"
python tools/ragged_decode_probe.py --small 2>&1 | grep "==\|DEVICE\|512 work\|768 work\|256 work\|EQUAL"
echo "== kbench rotating"; python tools/kbench.py decode --rotate --only "B16@32k,B8@32k,B4@32k,B64@8k,B256@2k,tp8 B64,B16@2k" 2>&1 | grep "splits="
