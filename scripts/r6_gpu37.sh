cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q -k "group or headline or parity" 2>&1 | grep -E "^FAILED|Error|assert|^E " | head -20
