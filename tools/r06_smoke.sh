cd "$(dirname "$0")/.."
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
( time python3 bench.py ) 2>&1 | tail -5 | cut -c1-400
