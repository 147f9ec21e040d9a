"""Stand-in for the five timm==0.9.2 names the reference hot path imports.

TEST INFRASTRUCTURE ONLY.  timm is not installed in this image and is not vendored by the
reference (requirements.txt:2 pins ``timm==0.9.2``; imports at
models/vision_transformer.py:8-9).  This package is OUR restatement of the public behaviour of
that release, written from its documented semantics; it exists so that
``oracle/pin_against_reference.py`` can execute the reference's OWN
``Attention/Block/CrossAttention/CrossBlock/VisionTransformerCustom`` classes on CPU and pin the
oracle against them.  Everything that lives here is therefore "parity unpinned" with respect to
timm itself (no copy of timm exists here to diff against) - see DESIGN.md section "Oracle".

It is never imported by the product package, by bench.py or by the GPU tests.
"""
