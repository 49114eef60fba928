"""oracle/ -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain torch fp32 ops, no HIP) of the ssUnet-GAN segmentation-GAN
training hot path, used as the checker for the HIP product path in `ssunet-gan_amd/`.

Rules (enforced by tests/test_layout_rules.py):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import anything from this package;
  * nothing here is shipped or measured as the product; the product path never falls
    back to it.

Pinning: every class/function here is checked against golden vectors that
`oracle/gen_golden.py` produced by importing the reference's own modules from
/root/reference in the build container (fixtures under tests/golden/).  The reference
repo holds no tests or golden vectors of its own, so those generated
fixtures are the only pin.
"""
