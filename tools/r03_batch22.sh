#!/bin/bash
# flake check: the whole GPU suite once more at HEAD in a fresh process
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -3
