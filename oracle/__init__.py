"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU (torch fp32 / numpy) restatement of the reference algorithm on the north-star path, used exclusively as the
checker by tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg.  Nothing under contrastors_amd/ may
import this package: the product path is the HIP library and fails loudly without it.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against fixtures under
tests/golden/ that were produced by importing the reference's own code (/root/reference/src/contrastors/loss.py and
models/huggingface/modeling_hf_nomic_bert.py) in the build container with oracle/make_golden.py.
"""
