"""opentransformer_amd -- MI355X-native hot path for otrans (placeholder, filled in below)."""
