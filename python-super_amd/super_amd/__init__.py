"""super_amd -- MI355X-native drop-in for SuPer's embedded-deformation LM hot path."""
