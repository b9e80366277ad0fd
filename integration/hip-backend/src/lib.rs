//! Safe Rust layer over libzkhip's C ABI (`include/zkhip.h`): the pieces the reference's `Prover`
//! (`crates/prover/src/prover/mod.rs`) needs from a GPU STARK engine -- a context per GPU, device buffers,
//! proving keys, `prove` / `verify` -- with errors mapped the way the reference maps SDK errors
//! (`Error::GenProof(String)`, mod.rs:357,378).
//!
//! SOURCE ONLY here (no Rust toolchain in the build image).  Wiring into the reference:
//! ```toml
//! # crates/prover/Cargo.toml, next to the `cuda` feature (:41-46)
//! hip = ["dep:openvm-hip-backend", "openvm-hip-backend/openvm-engine"]
//! ```
//! ```ignore
//! // crates/prover/src/prover/mod.rs:27-39, third arm of the engine aliases
//! #[cfg(feature = "hip")]
//! type DeferralEngine = openvm_hip_backend::engine::BabyBearPoseidon2HipEngine;
//! ```
pub mod ffi;
/// `BabyBearPoseidon2HipEngine`: the type `crates/prover` aliases behind the `hip` feature
/// (`crates/prover/src/prover/mod.rs:27-39`: `DeferralEngine`, `VerifyProver`, `VerifyCircuitProver`; constructed by
/// `<E as StarkEngine>::new(SystemParams)` at mod.rs:209).
///
/// SKETCH, not compiled: the trait surface below is recalled from the pinned `openvm-stark-backend` v2.0.0, whose
/// source is not vendored in the reference tree; the bodies state what each method hands to the C ABI.
/// Two gaps remain before this can replace the CUDA engine for real: the pinned backend's
/// protocol is the v2 sum-check/WHIR stack while libzkhip implements the v1 quotient + FRI pipeline the task names,
/// and `Proof<SC>`'s byte layout (`Encode`) is not this library's.
#[cfg(feature = "openvm-engine")]
pub mod engine {
    use crate::{AirDesc, HipContext, Params, ProvingKey};

    pub struct BabyBearPoseidon2HipEngine {
        pub ctx: HipContext,
        pub params: Params,
    }

    impl BabyBearPoseidon2HipEngine {
        /// `StarkEngine::new(params)` (mod.rs:209): one context on the current device.
        pub fn new(params: Params) -> Self {
            // upstream panics on a missing device (CudaError); keep that contract
            let ctx = HipContext::new(0).expect("no gfx950 device for the HIP engine");
            Self { ctx, params }
        }

        /// `StarkEngine::keygen` -> device proving key: every chip's symbolic constraints are lowered once to the
        /// bytecode of `AirDesc::program` (walk of `SymbolicConstraintsDag` nodes: Variable{Main|Preprocessed, offset} ->
        /// VAR/PREP, IsFirstRow/IsLastRow/IsTransition, Add/Sub/Mul/Neg, Constant, PublicValue; interactions ->
        /// the trailing section), then `zkhip_keygen`.
        pub fn keygen<'c>(&'c self, airs: &[AirDesc]) -> crate::Result<ProvingKey<'c>> {
            ProvingKey::keygen(&self.ctx, &self.params, airs)
        }

        /// `StarkEngine::prove(pk, ctx)`: traces are the device matrices the chips' trace generators produced
        /// (column-major Montgomery u32 -- the layout both backends use); public values go through `as_canonical_u32()`.
        pub fn prove(&self, pk: &ProvingKey<'_>, traces: &[&crate::DeviceBuffer<'_>], pvs: &[Vec<u32>]) -> crate::Result<Vec<u8>> {
            pk.prove(traces, pvs)
        }
    }
}

use std::ffi::CStr;
use std::marker::PhantomData;
use std::os::raw::c_void;
use std::ptr;

/// Mirrors `cudaError`-style statuses of the upstream backends: the negative code plus `zkhip_last_error`.
#[derive(Debug, Clone)]
pub struct HipError {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "zkhip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for HipError {}
pub type Result<T> = std::result::Result<T, HipError>;

/// One context per GPU and per `Prover`; not `Sync` (the reference proves one task at a time per `Prover`,
/// `gen_proof_universal(&mut self)`, mod.rs:287).  Different contexts may be used from different threads.
pub struct HipContext {
    raw: *mut ffi::zkhip_ctx,
}
unsafe impl Send for HipContext {}

impl HipContext {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        let rc = unsafe { ffi::zkhip_ctx_create(device, &mut raw) };
        if rc != ffi::ZKHIP_OK {
            return Err(HipError { code: rc, message: "no gfx950 device for the HIP backend".into() });
        }
        Ok(Self { raw })
    }
    fn check(&self, rc: i32) -> Result<()> {
        if rc == ffi::ZKHIP_OK {
            return Ok(());
        }
        let message = unsafe { CStr::from_ptr(ffi::zkhip_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(HipError { code: rc, message })
    }
    pub fn sync(&self) -> Result<()> {
        self.check(unsafe { ffi::zkhip_sync(self.raw) })
    }
    /// Device buffer of `len` BabyBear words (Montgomery form on the device, like p3's in-memory form).
    pub fn alloc(&self, len: usize) -> Result<DeviceBuffer<'_>> {
        let mut p: *mut c_void = ptr::null_mut();
        self.check(unsafe { ffi::zkhip_malloc(self.raw, len * 4, &mut p) })?;
        Ok(DeviceBuffer { ctx: self, ptr: p as *mut u32, len, _m: PhantomData })
    }
    /// Uploads Montgomery words as they are (p3 `BabyBear` values transmuted to `u32`).
    pub fn upload_monty(&self, words: &[u32]) -> Result<DeviceBuffer<'_>> {
        let b = self.alloc(words.len())?;
        self.check(unsafe { ffi::zkhip_h2d(self.raw, b.ptr as *mut c_void, words.as_ptr() as *const c_void, words.len() * 4) })?;
        Ok(b)
    }
    /// Trace of the Poseidon2 AIR for `n_perms` input states already on the device (row f3).
    pub fn poseidon2_air_tracegen(&self, inputs: &DeviceBuffer<'_>, n_perms: usize, log_height: u32, trace: &mut DeviceBuffer<'_>) -> Result<()> {
        assert!(inputs.len >= 16 * n_perms && trace.len >= ffi::ZKHIP_POSEIDON2_AIR_WIDTH << log_height);
        self.check(unsafe { ffi::zkhip_poseidon2_air_tracegen(self.raw, inputs.ptr, n_perms, log_height, trace.ptr) })
    }
    /// Multiplicity column of a range-check table from a requesting column that lies on the device.
    pub fn range_counts_tracegen(&self, values: &DeviceBuffer<'_>, n: usize, log_table: u32, counts: &mut DeviceBuffer<'_>, accumulate: bool) -> Result<()> {
        assert!(values.len >= n && counts.len >= 1usize << log_table);
        self.check(unsafe { ffi::zkhip_range_counts_tracegen(self.raw, values.ptr, n, log_table, counts.ptr, accumulate as i32) })
    }
}
impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { ffi::zkhip_ctx_destroy(self.raw) }
    }
}

/// Replaces `openvm_cuda_common::d_buffer::DeviceBuffer<F>`: owned device memory of one context.
pub struct DeviceBuffer<'c> {
    ctx: &'c HipContext,
    ptr: *mut u32,
    len: usize,
    _m: PhantomData<u32>,
}
impl<'c> DeviceBuffer<'c> {
    pub fn as_ptr(&self) -> *const u32 {
        self.ptr
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    pub fn download_monty(&self) -> Result<Vec<u32>> {
        let mut v = vec![0u32; self.len];
        self.ctx.check(unsafe { ffi::zkhip_d2h(self.ctx.raw, v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * 4) })?;
        Ok(v)
    }
}
impl Drop for DeviceBuffer<'_> {
    fn drop(&mut self) {
        unsafe { ffi::zkhip_free(self.ctx.raw, self.ptr as *mut c_void) };
    }
}

/// FRI / PoW parameters (`crates/circuits/*/openvm.toml:1-6`).
pub type Params = ffi::zkhip_params;
pub const CHUNK_CIRCUIT_PARAMS: Params =
    Params { log_blowup: 1, log_final_poly_len: 0, num_queries: 100, commit_pow_bits: 16, query_pow_bits: 16 };

/// One chip: its constraint bytecode (lowered from the chip's symbolic constraints), shape and
/// optional preprocessed trace.
pub struct AirDesc {
    pub program: Vec<u32>,
    pub log_height: u32,
    pub width: usize,
    pub n_pvs: usize,
    pub prep_trace: Option<Vec<u32>>,   // canonical, column-major: proving side
    pub prep_commit: Option<[u32; 8]>,  // verifying side
}
impl AirDesc {
    fn raw(&self) -> ffi::zkhip_air {
        ffi::zkhip_air {
            program: self.program.as_ptr(),
            program_len: self.program.len(),
            log_height: self.log_height,
            width: self.width,
            n_pvs: self.n_pvs,
            prep_trace: self.prep_trace.as_ref().map_or(ptr::null(), |v| v.as_ptr()),
            prep_commit: self.prep_commit.as_ref().map_or(ptr::null(), |v| v.as_ptr()),
        }
    }
}

/// Proving key of an AIR set for fixed trace heights: kernels compiled, workspace resident
/// (the counterpart of the device proving key the reference uploads once per `Sdk`, mod.rs:151-155).
pub struct ProvingKey<'c> {
    ctx: &'c HipContext,
    raw: *mut ffi::zkhip_pk,
    n_airs: usize,
}
impl<'c> ProvingKey<'c> {
    pub fn keygen(ctx: &'c HipContext, params: &Params, airs: &[AirDesc]) -> Result<Self> {
        let raws: Vec<ffi::zkhip_air> = airs.iter().map(AirDesc::raw).collect();
        let mut raw = ptr::null_mut();
        ctx.check(unsafe { ffi::zkhip_keygen(ctx.raw, params, raws.as_ptr(), raws.len(), &mut raw) })?;
        Ok(Self { ctx, raw, n_airs: airs.len() })
    }
    pub fn proof_size(&self) -> usize {
        unsafe { ffi::zkhip_proof_size(self.raw) }
    }
    pub fn prep_commitment(&self, air_index: usize) -> Result<[u32; 8]> {
        let mut out = [0u32; 8];
        self.ctx.check(unsafe { ffi::zkhip_pk_prep_commitment(self.ctx.raw, self.raw, air_index, out.as_mut_ptr()) })?;
        Ok(out)
    }
    /// `traces[i]`: device, column-major, Montgomery, stride `1 << log_height`; `pvs[i]`: canonical public values.
    pub fn prove(&self, traces: &[&DeviceBuffer<'_>], pvs: &[Vec<u32>]) -> Result<Vec<u8>> {
        assert!(traces.len() == self.n_airs && pvs.len() == self.n_airs);
        let tp: Vec<*const u32> = traces.iter().map(|t| t.as_ptr()).collect();
        let pp: Vec<*const u32> = pvs.iter().map(|p| p.as_ptr()).collect();
        let mut out = vec![0u8; self.proof_size()];
        let mut len = 0usize;
        self.ctx.check(unsafe { ffi::zkhip_prove(self.ctx.raw, self.raw, tp.as_ptr(), pp.as_ptr(), out.as_mut_ptr(), out.len(), &mut len) })?;
        out.truncate(len);
        Ok(out)
    }
}
impl Drop for ProvingKey<'_> {
    fn drop(&mut self) {
        unsafe { ffi::zkhip_pk_destroy(self.ctx.raw, self.raw) }
    }
}

/// Host verifier (no device): what `UniversalVerifier::verify_stark_proof_with_vk` calls in place of
/// `Sdk::verify_proof` (crates/verifier/src/verifier.rs:82) for this backend's proofs.
pub fn verify(params: &Params, airs: &[AirDesc], pvs: &[Vec<u32>], proof: &[u8]) -> std::result::Result<(), i32> {
    let raws: Vec<ffi::zkhip_air> = airs.iter().map(AirDesc::raw).collect();
    let pp: Vec<*const u32> = pvs.iter().map(|p| p.as_ptr()).collect();
    match unsafe { ffi::zkhip_verify(params, raws.as_ptr(), raws.len(), pp.as_ptr(), proof.as_ptr(), proof.len()) } {
        ffi::ZKHIP_OK => Ok(()),
        rc => Err(rc),
    }
}
