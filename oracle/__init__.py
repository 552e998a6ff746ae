"""CPU oracle of the LSPIV hot path -- test infrastructure, never imported by the product (pyorc_amd)."""
