"""TEST INFRASTRUCTURE - CPU oracle of the Bi_Diffusion hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; nothing under the product package does. It restates the reference's algorithms on
the CPU (PyTorch fp32 for the floating-point denoiser, plain C for the integer sampling
logic) and is pinned against golden vectors captured from the imported reference
(tests/golden/).
"""
