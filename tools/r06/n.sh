#!/bin/bash
# round 6, call n: who issues AliNet's copyBuffer launches
python tools/_exp/alinet_copies.py 2>&1 | tail -32
