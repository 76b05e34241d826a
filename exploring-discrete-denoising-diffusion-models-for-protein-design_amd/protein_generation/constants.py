"""Alphabets of the protein-style process (protein_generation/constants.py:1-10).
Token order is part of the data format: index = position in PROTEIN_ALPHABET."""
CAN_AAS = "".join(("ACDEFGHIKL", "MNPQRSTVWY"))   # 20 canonical amino acids
AMB_AAS = "X"                                      # ambiguous residue
GAP, MASK, MSA_PAD = "-", "#", "!"

ALL_AAS = CAN_AAS + AMB_AAS                        # 21
MSA_AAS = ALL_AAS + GAP                            # 22 = K of the diffusion
PROTEIN_ALPHABET = MSA_AAS + MASK + MSA_PAD        # 24 = widest denoiser output
