"""Test infrastructure: CPU restatement of the reference algorithm (see restated.py). Not part of the product."""
