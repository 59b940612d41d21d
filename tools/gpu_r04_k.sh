#!/bin/bash
timeout 600 python3 -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
