"""pytheiasfm_amd -- MI355X-native bundle adjustment + RANSAC engine behind
pyTheia's BundleAdjust* / Estimate* entry points."""
__version__ = "0.1.0"
