// Links libzkhip.so (built by `python -c "import __graft_entry__ as g; g.build()"` or `make -C zkvm-prover_amd/csrc`).
fn main() {
    let dir = std::env::var("ZKHIP_LIB_DIR").expect("set ZKHIP_LIB_DIR to the directory holding libzkhip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=ZKHIP_LIB_DIR");
}
