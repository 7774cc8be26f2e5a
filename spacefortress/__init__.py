"""Drop-in namespace for the reference's `spacefortress` packages (`import spacefortress.gym`,
`import spacefortress.core as sf`), backed by the MI355X engine in `spacefortress_amd`."""
